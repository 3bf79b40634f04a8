// tuning.h -- where measurement switches live (internal header).
//
// Two kinds of environment variables reach this library:
//   * behaviour the tests and callers rely on (cross-check paths, layout choices, thread counts): read with
//     std::getenv where they are used -- NMFAMD_FORCE_VALU, NMFAMD_NO_FUSED_MU, NMFAMD_ONE_IMAGE, NMFAMD_FP_TILE,
//     NMFAMD_GRAM_PARTIALS, NMFAMD_GRAM_KSPLIT, NMFAMD_X3_COLSPLIT, NMFAMD_TRI_RIDE, NMFAMD_ERROR_MEMCPY, NMFAMD_SHARD_REHEARSE, NMFAMD_SHARD_NO_DIRECT, NMFAMD_HOST_THREADS, NMFAMD_COMM, NMFAMD_MALL_MB, NMFAMD_ONE_PASS, NMFAMD_KL_BLOCK_KB, NMFAMD_TRI_FP32_DEN;
//   * A/B switches and stamped kernel variants that exist for measurements only: those go through tuning_env() and are
//     dead code in the shipped library -- they are compiled in by `python -m nmfgpu_amd.build --diag`
//     (-DNMFAMD_DIAG_BUILD, output lib/libnmfgpu64_diag.so; select it with NMFAMD_LIBRARY).
#pragma once

#include <cstdlib>

namespace nmfamd {

#ifdef NMFAMD_DIAG_BUILD
constexpr bool DIAG_BUILD = true;
inline const char* tuning_env(const char* name) { return std::getenv(name); }
#else
constexpr bool DIAG_BUILD = false;
inline const char* tuning_env(const char*) { return nullptr; }
#endif

} // namespace nmfamd
