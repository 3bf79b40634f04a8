// tuning.h -- where measurement switches live (internal header).
//
// Two kinds of environment variables reach this library:
//   * behaviour callers may rely on, read with std::getenv where they are used -- the complete list (tests/test_abi.py checks the shipped .so's strings against it):
//       NMFAMD_COMM (transport of a numGpus team: rccl / p2p), NMFAMD_SELFTEST (0: skip the peer transport's set-up self-test), NMFAMD_HOST_THREADS (host initialisers),
//       NMFAMD_MALL_MB (size of the memory-side cache when the device does not report it), NMFAMD_KL_BLOCK_KB (L2 block of the KL gather), NMFAMD_ONE_IMAGE (0 / 1: two
//       images of V or one), and the cross-check PATHS the parity tests compare with each other -- all of them
//       complete, correct implementations: NMFAMD_FORCE_VALU, NMFAMD_NO_FUSED_MU (fp32 rank 64 and, round 6, double precision: the generic launch sequence),
//       NMFAMD_GRAM_PARTIALS, NMFAMD_FP_TILE, NMFAMD_SPARSE_SETUP (host: the sparse images are built by the host path, the device path's fall-back);
//   * A/B switches, forced kernel forms, rehearsal modes and stamped kernel variants that exist for measurements and form-against-form tests only: those go through
//     tuning_env() and are dead code in the shipped library -- they are compiled in by `python -m nmfgpu_amd.build --diag` (-DNMFAMD_DIAG_BUILD, output
//     lib/libnmfgpu64_diag.so; tests/conftest.py `diag_build`, tools: NMFAMD_LIBRARY).  Round 5 moved here: NMFAMD_SHARD_REHEARSE (a value > 1 makes a rank update
//     1 / N of W's rows: timing only, the factors mean nothing), NMFAMD_SHARD_NO_DIRECT, NMFAMD_ERROR_MEMCPY, NMFAMD_GRAM_KSPLIT, NMFAMD_X3_COLSPLIT, NMFAMD_TRI_RIDE,
//     NMFAMD_TRI_FP32_DEN, NMFAMD_NORMALIZE_TWO_LAUNCHES; also NMFAMD_ERROR_COPY_KERNEL (error terms through k_copy_small instead of written by the update kernels) and
//     NMFAMD_GRAM_SPREAD (0: a tile per Gram passenger instead of the spread form).  Round 6 moved here: NMFAMD_ONE_PASS with its kernel (kernels_onepass.hip is
//     compiled into the measurement build only), and added NMFAMD_RIDE64_STOP / NMFAMD_F64_STAMPS / NMFAMD_F64_MIN_STEPS / NMFAMD_F64_HALF_TILES /
//     NMFAMD_RIDE64_PID_ORDER (double-precision fused iteration), NMFAMD_BF_VARIANT (the config-4 product's loop with parts taken out) and NMFAMD_F32W_RIDE
//     (0 / 1: the wide fp32 Gram slices never / always as passengers of the product launch).
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
