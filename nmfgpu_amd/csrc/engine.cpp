// engine.cpp -- device-resident factorisation state and the per-iteration kernel sequences.
//
// Operation order per algorithm follows the reference's computeIteration bodies:
//   MU     source/nmf/AlgorithmMultiplicativeFrobenius.h:150-248
//   nsNMF  source/nmf/AlgorithmNonSmoothNMF.h:159-226
//   GDCLS  source/nmf/AlgorithmGradientDescentConstrainedLeastSquares.h:159-272
//   ACLS / AHCLS  source/nmf/AlgorithmAlternatingHoyerConstrainedLeastSquares.h:171-296
//   ALS    source/nmf/AlgorithmAlternatingLeastSquares.h:146-224
// but the eight vendor-BLAS calls + five custom kernels of one iteration collapse into the
// fused kernels of kernels.hip (see DESIGN.md section 4 for the mapping).
#include "engine.h"
#include "tuning.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>

namespace nmfamd {

#define HIPX(call)                                        \
	do {                                                  \
		hipError_t e__ = (call);                          \
		if (e__ != hipSuccess) return hip_fail(e__, #call); \
	} while (0)

template <typename T>
double resolve_frobenius(const std::vector<T>& vtv_sorted, std::vector<T>& htwtv, std::vector<T>& hhtwtw) {
	return std::sqrt(resolve_frobenius_squared<T>(vtv_sorted, htwtv, hhtwtw));
}
template double resolve_frobenius<float>(const std::vector<float>&, std::vector<float>&, std::vector<float>&);
template double resolve_frobenius<double>(const std::vector<double>&, std::vector<double>&, std::vector<double>&);

// Ascending sort of the error-term vectors (std::sort in source/nmf/FrobeniusResolver.cpp:33-35).  A sorted sequence of values is the same whatever the
// algorithm, so long vectors go through a radix sort on the order-preserving integer image of the floating-point bits (three or four passes over n keys instead of
// n log n comparisons): at config 2's n = 5 000 the comparison sort cost the drop-in loop ~0.2 ms of host time per error iteration -- with the device idle, because
// whether the run goes on depends on the value (nmfgpu::compute, thresholdValue) -- 108 us per iteration end to end against 87 in the resident engine.
// NaNs (a diverged run) keep std::sort's company: they are sent to the comparison sort.
template <typename T>
static void sort_terms(std::vector<T>& v) {
	const size_t n = v.size();
	if (n < 256) { std::sort(v.begin(), v.end()); return; }
	for (size_t i = 0; i < n; ++i) if (v[i] != v[i]) { std::sort(v.begin(), v.end()); return; }
	using U = typename std::conditional<sizeof(T) == 4, std::uint32_t, std::uint64_t>::type;
	constexpr int BITS = 11, BUCKETS = 1 << BITS, PASSES = (int)((8 * sizeof(T) + BITS - 1) / BITS);
	constexpr U SIGN = (U)1 << (8 * sizeof(T) - 1);
	std::vector<U> a(n), b(n);
	for (size_t i = 0; i < n; ++i) {
		U u;
		std::memcpy(&u, &v[i], sizeof(T));
		a[i] = (u & SIGN) ? ~u : (u | SIGN);          // negative values: all bits flipped; others: sign bit set -- unsigned order = numeric order (-0 before +0)
	}
	std::vector<size_t> count(BUCKETS);
	for (int p = 0; p < PASSES; ++p) {
		const int shift = p * BITS;
		std::fill(count.begin(), count.end(), (size_t)0);
		for (size_t i = 0; i < n; ++i) ++count[(a[i] >> shift) & (BUCKETS - 1)];
		size_t run = 0;
		for (int k = 0; k < BUCKETS; ++k) { const size_t c = count[k]; count[k] = run; run += c; }
		for (size_t i = 0; i < n; ++i) b[count[(a[i] >> shift) & (BUCKETS - 1)]++] = a[i];
		a.swap(b);
	}
	for (size_t i = 0; i < n; ++i) {
		const U u = (a[i] & SIGN) ? (a[i] & ~SIGN) : ~a[i];
		std::memcpy(&v[i], &u, sizeof(T));
	}
}

template <typename T>
double resolve_frobenius_squared(const std::vector<T>& vtv_sorted, std::vector<T>& htwtv, std::vector<T>& hhtwtw) {
	sort_terms(htwtv);
	sort_terms(hhtwtw);
	double acc = 0.0;
	const size_t mx = std::max(vtv_sorted.size(), std::max(htwtv.size(), hhtwtw.size()));
	for (size_t j = 0; j < mx; ++j) {
		if (j < vtv_sorted.size()) acc += vtv_sorted[j];
		if (j < htwtv.size()) acc -= 2.f * htwtv[j];   // float literal: the product is formed in T, as in the reference
		if (j < hhtwtw.size()) acc += hhtwtw[j];
	}
	return acc;
}
template double resolve_frobenius_squared<float>(const std::vector<float>&, std::vector<float>&, std::vector<float>&);
template double resolve_frobenius_squared<double>(const std::vector<double>&, std::vector<double>&, std::vector<double>&);

// Size of the device's memory-side cache (AMD Infinity Cache): not in hipDeviceProp_t, so a table by architecture -- 256 MiB
// on gfx942 / gfx950 in SPX mode (/opt/skills/guides/MI355X_MICROARCH.md, "Infinity Cache (L3) 256 MiB") -- which
// NMFAMD_MALL_MB overrides (other parts, partitioned modes, several engines sharing the cache).  0 = unknown: no window.
static size_t memory_side_cache_bytes(const hipDeviceProp_t& prop) {
	if (const char* e = std::getenv("NMFAMD_MALL_MB")) { const long mb = std::atol(e); return mb > 0 ? (size_t)mb << 20 : 0; }
	if (std::strncmp(prop.gcnArchName, "gfx942", 6) == 0 || std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) return (size_t)256 << 20;
	return 0;
}

template <typename T>
Engine<T>::Engine(int m, int n, int r, int algorithm, const AlgorithmParams& params)
	: m_(m), n_(n), r_(r), RP_(padded_rank(r, (params.sparse_compute != 0 || params.divergence != 0) ? 4 : sizeof(T))), alg_(algorithm), prm_(params), mpad_(pad128(m)), npad_(pad128(n)) {}

template <typename T>
Status Engine<T>::hip_fail(hipError_t e, const char* what) {
	last_error_ = what;
	(void)hipGetLastError();
	return e == hipErrorOutOfMemory ? ST_NO_DEVICE_MEMORY : ST_HIP_ERROR;
}

#ifdef NMFAMD_DIAG_BUILD
// measurement build, NMFAMD_BF_STAMPS=<file>: the waves' life stamps of the last W^T V (kind 0) and V H^T (kind 1) launch of the rank-256 bf16 product, written
// to the file when an engine is destroyed (tools/stamp_bf16.py): [kind][512 workgroups][4 waves][entry, loop start, loop end, exit] in 100 MHz ticks
static unsigned long long* g_bf_stamps = nullptr;
static unsigned long long* bf_stamps(int kind) {
	static const char* path = std::getenv("NMFAMD_BF_STAMPS");
	if (path == nullptr) return nullptr;
	if (g_bf_stamps == nullptr) {
		if (hipMalloc((void**)&g_bf_stamps, 2 * 512 * 16 * sizeof(unsigned long long)) != hipSuccess) { g_bf_stamps = nullptr; return nullptr; }
		(void)hipMemset(g_bf_stamps, 0, 2 * 512 * 16 * sizeof(unsigned long long));
	}
	return g_bf_stamps + (long)kind * 512 * 16;
}
static void bf_stamps_dump() {
	const char* path = std::getenv("NMFAMD_BF_STAMPS");
	if (g_bf_stamps == nullptr || path == nullptr) return;
	std::vector<unsigned long long> h(2 * 512 * 16);
	(void)hipDeviceSynchronize();
	if (hipMemcpy(h.data(), g_bf_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
		if (FILE* f = std::fopen(path, "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); }
	}
}
#endif

template <typename T>
Engine<T>::~Engine() {
#ifdef NMFAMD_DIAG_BUILD
	if (bf16_) bf_stamps_dump();
#endif
	T* bufs[] = {V_, Vt_, Wt_, H_, Ws_, Hs_, slabs_, numW_, Wold_, G_, G2_, HHt_, Qinv_, gram_part_, sumsq_part_, psN_, stage_};   // (psR_ lives behind psN_)
	for (T* b : bufs) if (b) (void)hipFree(b);
	if (inv_work_) (void)hipFree(inv_work_);
	if (range_flag_) (void)hipFree(range_flag_);
	{
		void* sp[] = {csr_ptr_, csr_idx_, csc_ptr_, csc_idx_, csc_from_csr_, csr_val_, csc_val_, q_, q2_, t_vwh_, t_kl_, rowsum_part_, sW_, sH_, kl_scale_, csr_bptr_, csc_bptr_, kl_part_, kl_tpart_};
		for (void* b : sp) if (b) (void)hipFree(b);
	}
	{ void* bb[] = {Vb_, Vtb_, Wtb_, Hb_, Wx3_, Hx3_, qx3_, gram_tri_part_, Gw_raw_, Gh_raw_, colsq_}; for (void* b : bb) if (b) (void)hipFree(b); }
	if (f64_stamps_ != nullptr) {
		std::vector<unsigned long long> h((size_t)4 * 4096 * 8);
		if (hipMemcpy(h.data(), f64_stamps_, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
			if (FILE* f = std::fopen(tuning_env("NMFAMD_F64_STAMPS"), "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); }
		}
		(void)hipFree(f64_stamps_);
	}
	if (f32w_scale_) (void)hipFree(f32w_scale_);
	{ void* fb[] = {f64_scale_, f64_partial_, f64_counters_, f64_items_h_, f64_items_w_}; for (void* b : fb) if (b) (void)hipFree(b); }
	if (gramW_part_) (void)hipFree(gramW_part_);
	if (wsq_part_) (void)hipFree(wsq_part_);
	if (rowdot_part_) (void)hipFree(rowdot_part_);
	if (tri_ride_counters_) (void)hipFree(tri_ride_counters_);
	if (Gpart_) (void)hipFree(Gpart_);
	if (gram_spread_counter_) (void)hipFree(gram_spread_counter_);
	if (Graw64_) (void)hipFree(Graw64_);
	if (gramH_part_) (void)hipFree(gramH_part_);
	if (scale_) (void)hipFree(scale_);
	{ void* ob[] = {op_part_, op_hfrag_, op_ctl_, op_slabs_, op_hh_part_, op_ps4_, op_H2_, op_stamps_}; for (void* b : ob) if (b) (void)hipFree(b); }
	if (pin_abort_) (void)hipHostFree(pin_abort_);
	if (err_event_) (void)hipEventDestroy(err_event_);
	if (ev_fork_) (void)hipEventDestroy(ev_fork_);
	if (ev_join_) (void)hipEventDestroy(ev_join_);
	if (aux_) (void)hipStreamDestroy(aux_);
	if (pin_psN_) (void)hipHostFree(pin_psN_);   // (pin_psR_ lives behind it)
	if (pin_kl_) (void)hipHostFree(pin_kl_);
	for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
}

template <typename T>
Status Engine<T>::allocate() {
	if (m_ <= 0 || n_ <= 0 || r_ <= 0 || alg_ < 0 || alg_ > ALG_NSNMF) return ST_INVALID;
	int dev = 0;
	HIPX(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIPX(hipGetDeviceProperties(&prop, dev));
	num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

	// both products run on the MFMA pipe: fp32 (kernels.hip) or fp64 (kernels_f64.hip), each with its own cut
	const bool f64 = std::is_same<T, double>::value;
	planH_ = f64 ? plan_factor_product_f64(n_, m_, RP_, num_cus_) : plan_factor_product(n_, m_, RP_, num_cus_);
	planW_ = f64 ? plan_factor_product_f64(m_, n_, RP_, num_cus_) : plan_factor_product(m_, n_, RP_, num_cus_);
	const bool mfma = std::getenv("NMFAMD_FORCE_VALU") == nullptr;
	// ranks <= 32: the fp32 product computes 32 panel columns instead of 64 (half the MFMA work; the kernel turns HBM-bound)
	if (RP_ == 64 && r_ <= 32 && tuning_env("NMFAMD_FP_FULL_WIDTH") == nullptr) { planH_.nb = f64 ? 2 : 1; planW_.nb = planH_.nb; }      // (fp64 counts 16-column tiles)
	if (!mfma) { planH_.splits = 1; planW_.splits = 1; planH_.th = planW_.th = 128; planH_.xtiles = (int)(pad128(n_) / 128); planW_.xtiles = (int)(pad128(m_) / 128); }
	tiled_ = mfma;
	sparse_ = prm_.sparse_compute != 0 || prm_.divergence != 0;      // (the SpMM / KL kernels gather RP / 64 = 1, 2 or 4 values per lane: sparse runs keep the 128-column padding in either precision, see the constructor)
	if (sparse_) {
		if (alg_ != ALG_MU || RP_ > 256) return ST_INVALID;   // sparse compute: multiplicative update, padded rank 64 / 128 / 256
		tiled_ = false;
		planH_.splits = planW_.splits = 1; planH_.th = planW_.th = 128;
		planH_.xtiles = (int)(pad128(n_) / 128); planW_.xtiles = (int)(pad128(m_) / 128);
	}
	if (prm_.precision > 0) {
		// bf16 operands for the two big products, dense (resident) V only; every algorithm, padded rank 64 or k * 128
		if (!std::is_same<T, float>::value || sparse_ || !mfma) return ST_INVALID;
		bf16_ = true;
		tiled_ = false;
		planH_.th = planW_.th = 128;
		planH_.xtiles = (int)(pad128(n_) / 128); planW_.xtiles = (int)(pad128(m_) / 128);
		ksW_ = (n_ + 15) / 16; ksH_ = (m_ + 15) / 16;
		planH_.splits = plan_splits_bf16(planH_.xtiles, ksH_, RP_, num_cus_);
		planW_.splits = plan_splits_bf16(planW_.xtiles, ksW_, RP_, num_cus_);
		// Padded rank 256 (the kernels_tri.hip iteration): the Gram matrix of the operand a product multiplies V with rides in that product's launch as TRI_PASSENGERS
		// workgroups (tri_gram_tile.h) -- its consumer is the update kernel behind the launch.  V (S H)^T leaves the CUs free as it is (config 4: 224 workgroups);
		// W^T V is planned with one K slice fewer (9 -> 8: 252 -> 224 workgroups).  NMFAMD_TRI_RIDE = 0 / h / w: none / only (S H)(S H)^T / only W^T W.
		if (RP_ == 256 && (alg_ == ALG_MU || alg_ == ALG_NSNMF) && tri_kernels_available(RP_)) {
			const char* e = tuning_env("NMFAMD_TRI_RIDE");
			const bool want_w = e == nullptr || (e[0] != '0' && e[0] != 'h'), want_h = e == nullptr || (e[0] != '0' && e[0] != 'w');
			if (want_w && num_cus_ > 2 * TRI_PASSENGERS) {
				FactorProductPlan t = planH_;
				t.splits = plan_splits_bf16(planH_.xtiles, ksH_, RP_, num_cus_ - TRI_PASSENGERS);
				if (bf16_product_workgroups(t) + TRI_PASSENGERS <= num_cus_) { planH_.splits = t.splits; tri_ride_w_ = true; }
			}
			tri_ride_h_ = want_h && bf16_product_workgroups(planW_) + TRI_PASSENGERS <= num_cus_;
		}
		planHb_ = planH_; planWb_ = planW_;
	}
	// fp32, dense, MFMA path: both products run on the bf16 matrix pipe with every operand split exactly into
	// three bf16 terms (fp32-level accuracy, six cross products; kernels_x3.hip) -- HBM-bound instead of bound by
	// the fp32 MFMA rate.  precision = -1 (or NMFAMD_FP32_NATIVE) keeps the native fp32 MFMA instructions.
	// (ranks <= 32 stay on the fp32 MFMA kernel: with half its MFMA work it is HBM-bound already and needs no split image)
	// ... except where one image in the memory-side cache pays more than the halved MFMA work (§3: 107 -> 100 us at config 2's shape)
	// One resident image pays when it fits the memory-side (Infinity) cache and two do not: a window around the cache size,
	// computed once (cache_window_) from the device (memory_side_cache_bytes)
	const size_t image_bytes = sizeof(T) * (size_t)pad128(m_) * (size_t)pad128(n_);
	const size_t mall = memory_side_cache_bytes(prop);
	const bool cache_window = mall > 0 && (double)image_bytes > 0.6 * (double)mall && (double)image_bytes < 1.12 * (double)mall && std::getenv("NMFAMD_ONE_IMAGE") == nullptr;
	// (the opt-in one-pass iteration runs on the split-operand products whatever the rank)
#ifdef NMFAMD_DIAG_BUILD
	const char* op_env = tuning_env("NMFAMD_ONE_PASS");
	const bool op_req = op_env != nullptr && std::atoi(op_env) != 0 && one_pass_allowed_ && row_blocks_ == 1 && fused_capable() && RP_ == 64 &&
	                    std::getenv("NMFAMD_GRAM_PARTIALS") == nullptr && tuning_env("NMFAMD_IMAGE_TILE128") == nullptr && onepass_available(pad128(m_), num_cus_);
#else
	const bool op_req = false;      // (kernels_onepass.hip is part of the measurement build only: round 6)
#endif
	if (std::is_same<T, float>::value && tiled_ && !bf16_ && !sparse_ && RP_ % 64 == 0 && prm_.precision == 0 && (planH_.nb == 2 || cache_window || op_req) &&
	    tuning_env("NMFAMD_FP32_NATIVE") == nullptr) {
		planH_.nb = planW_.nb = 2;
		x3_ = true;
		planH_.th = planW_.th = 128;
		planH_.xtiles = (int)(pad128(n_) / 128); planW_.xtiles = (int)(pad128(m_) / 128);
		ksW_ = (n_ + 15) / 16; ksH_ = (m_ + 15) / 16;
		// W^T W rides in the W^T V launch as passenger workgroups (gram_image.h).  Its chain -- the fragments of all m rows through ONE CU per tile, 24 us at
		// config 2's m -- must stay shorter than the product: for a column shard narrower than config 2 the K range is
		// cut into slices (10 passenger workgroups each, CUs the product plan leaves free); config 2 itself keeps one slice and its plan (measured hidden).
		gram_ksplit_ = 1;
		if (fused_capable() && RP_ == 64 && std::getenv("NMFAMD_GRAM_PARTIALS") == nullptr) {
			const double product_us = 6.0 + (double)sizeof(T) * (double)pad128(m_) * (double)pad128(n_) / 5.0e6;       // (bytes at ~5 TB/s + launch and fill)
			const double pairs = (ksH_ + 2) / 2;
			while (gram_ksplit_ < GRAM_KSPLIT_MAX && 24.0 * (pairs / 317.0) / gram_ksplit_ > 0.55 * product_us) gram_ksplit_ *= 2;      // (24 us for config 2's 317 pairs in one slice: one CU's L2 rate, gram_image.h)
			// (NMFAMD_GRAM_KSPLIT = 1 / 2 / 4 / 8 forces the slice count: the parity tests run every form at shapes the oracle covers)
			if (const char* e = tuning_env("NMFAMD_GRAM_KSPLIT")) { const int k = std::atoi(e); if (k == 1 || k == 2 || k == 4 || k == 8) gram_ksplit_ = k; }
		}
		planH_.splits = plan_splits_x3(planH_.xtiles, ksH_, num_cus_, gram_ksplit_ > 1 ? GRAM_IMAGE_TILES * gram_ksplit_ : 0);
		planW_.splits = plan_splits_x3(planW_.xtiles, ksW_, num_cus_);
		planHx_ = planH_; planWx_ = planW_;
		planHx_.steps_total = ksH_; planWx_.steps_total = ksW_;
		planHx_.nb = planWx_.nb = 2;
		// One resident image of V or two?  With two, each product streams its own image along its output index (the
		// faster kernel form); with one, W^T V reads the image of V along its reduction index (y-tiled form, ~20 % slower
		// per launch on its own).  One image wins when it fits the 256 MiB memory-side cache but two do not: both products
		// of an iteration then find most of V there (config 2: V H^T 49 -> 37 us, iteration 126 -> 108 us).  Measured
		// (10 000 x 5 000 scaled, r = 64): 148 MB two images 97 us / one 102; 185 MB 109 / 100; 207 MB 126 / 108; 246 MB
		// 145 / 129; 266 MB 152 / 143; 328 MB 173 / 174; 1.6 GB 792 / 988.  Also one image when two (plus the staging
		// image of the upload) do not fit in HBM.  NMFAMD_ONE_IMAGE = 1 / 0 forces the choice.
		size_t free_b = 0, total_b = 0;
		const size_t image_b = sizeof(T) * (size_t)pad128(m_) * (size_t)pad128(n_);
		const char* force = std::getenv("NMFAMD_ONE_IMAGE");
		// the one-pass iteration needs ONE image (16-row tiles) and nothing else
		// (measurement build, NMFAMD_ONE_PASS=1: measured slower than the two-pass iteration, docs/HISTORY.md)
		one_pass_ = op_req && (force == nullptr || std::atoi(force) != 0);
		if (one_pass_) one_image_ = true;
		else if (force != nullptr) one_image_ = std::atoi(force) != 0;
		else if (cache_window) one_image_ = true;
		else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && 3 * image_b + (image_b >> 3) > free_b) one_image_ = true;
		// V H^T of a narrow column shard (short reduction range: at most 20 K-steps per wave without K slices): ONE slab from 128 x 32 workgroups, two per x-tile and two to a
		// CU (kernels_x3.hip, NBW = 1), instead of up to three K slices of 128 x 64 ones.  Per wave the 128 x 64 form is bound by its own MFMA issue (35 cycles each) while
		// a third of the chip's SIMDs work; the narrow form puts a second wave on every SIMD (the operand split is done twice, so it pays only while the range is short).
		// Same bits as the 128 x 64 form with one slice; and one slab is the exchange panel of the sharded loop as it stands (no k_reduce_slabs launch).
		// Measured at m = 10 000 (tools/shard_trace.py; fused loop / rank-of-N rehearsal, us per iteration): n = 625: 39.4 -> 36.6 / 40.4 -> 37.5; n = 1 250: 44.4 -> 44.3 /
		// 48.5 -> 45.3; n = 1 900: 50.8 -> 52.9 / 54.8 -> 54.0; n = 2 500: 58.5 -> 61.7 / 63.4 -> 62.5 -- hence the limit of 80 K-steps.  Both products in this form at
		// config 2: 92.7 -> 106.7 us (NMFAMD_X3_COLSPLIT = 2).  NMFAMD_X3_COLSPLIT = 0 / 1 forces the choice (parity test, measurements).
		w_col_split_ = false;
		if (RP_ == 64 && fused_capable() && !one_image_ && 2 * planW_.xtiles + GRAM_REDUCE_BLOCKS <= 2 * num_cus_) {
			w_col_split_ = ksW_ <= 80 && 4 * planW_.xtiles >= num_cus_;
			if (const char* e = tuning_env("NMFAMD_X3_COLSPLIT")) { const int v = std::atoi(e); if (v == 0) w_col_split_ = false; else if (v == 1) w_col_split_ = true; }
			if (w_col_split_) { planW_.splits = planWx_.splits = 1; planWx_.col_split = 2; }
		}
		// (measurement builds, NMFAMD_X3_COLSPLIT = 2: both products from 128 x 32 workgroups, two per CU, with the plan's K slices)
		if (const char* e = tuning_env("NMFAMD_X3_COLSPLIT")) { if (std::atoi(e) == 2 && RP_ == 64 && fused_capable()) { planHx_.col_split = planWx_.col_split = 2; w_col_split_ = true; } }
	}
	// panels (and the slabs the products write) cover whole x-tiles and whole 128-column update tiles
	mpad_ = pad128(std::max<long>(m_, (long)planW_.xtiles * planW_.th));
	if (row_blocks_ > 1) { const long g = 128l * row_blocks_; mpad_ = ((mpad_ + g - 1) / g) * g; }   // equal row blocks of whole 128-row tiles
	npad_ = pad128(std::max<long>(n_, (long)planH_.xtiles * planH_.th));
	img_th_ = (one_image_ && tuning_env("NMFAMD_IMAGE_TILE128") == nullptr) ? 16 : planW_.th;
	strideV_ = (long)img_th_ * npad_;
	strideVt_ = (long)planH_.th * mpad_;
	elemsV_ = mpad_ * npad_;     // tiled or not: every tile spans all columns
	elemsVt_ = tiled_ ? (long)planH_.xtiles * strideVt_ : mpad_ * npad_;
	slab_stride_ = (long)RP_ * std::max(mpad_, npad_);
	const long slab_elems = slab_stride_ * std::max(planH_.splits, planW_.splits);
	const long panelW = (long)RP_ * mpad_, panelH = (long)RP_ * npad_, rr = (long)RP_ * RP_;

	auto dalloc = [&](T** p, long elems) -> hipError_t {
		hipError_t e = hipMalloc((void**)p, (size_t)elems * sizeof(T));
		if (e != hipSuccess) return e;
		return hipMemsetAsync(*p, 0, (size_t)elems * sizeof(T), stream_);
	};
	if (bf16_) {
		HIPX(hipMalloc(&Vb_, 16 * (size_t)planW_.xtiles * ksW_ * 256));
		HIPX(hipMalloc(&Vtb_, 16 * (size_t)planH_.xtiles * ksH_ * 256));
		{
			// (row-block sharded runs all-gather the fragments in place, world x blk_rows / 16 K-steps = mpad_ / 16 of them -- more than ksH_ whenever m is not a
			//  multiple of 128 x world (config 4's 50 000 rows on 8 ranks are padded to 50 176); the K-steps behind ksH_ are never read by a product and stay zero)
			const size_t ks_alloc = (size_t)std::max<long>(ksH_, mpad_ / 16);
			HIPX(hipMalloc(&Wtb_, 16 * ks_alloc * (RP_ / 32) * 64));
			if (ks_alloc > (size_t)ksH_) HIPX(hipMemsetAsync((char*)Wtb_ + 16 * (size_t)ksH_ * (RP_ / 32) * 64, 0, 16 * (ks_alloc - (size_t)ksH_) * (RP_ / 32) * 64, stream_));
		}
		HIPX(hipMalloc(&Hb_, 16 * (size_t)ksW_ * (RP_ / 32) * 64));
	} else if (!sparse_) {
		if (x3_) {
			// + 1: the all-zero K-step that closes the image (written here once; the update kernels never touch it)
			const size_t bw = 3 * 16 * (size_t)(ksH_ + 1) * (RP_ / 32) * 64, bh = 3 * 16 * (size_t)(ksW_ + 1) * (RP_ / 32) * 64;
			HIPX(hipMalloc(&Wx3_, bw));
			HIPX(hipMalloc(&Hx3_, bh));
			HIPX(hipMemsetAsync(Wx3_, 0, bw, stream_));
			HIPX(hipMemsetAsync(Hx3_, 0, bh, stream_));
		}
		HIPX(dalloc(&V_, elemsV_));
		if (!one_image_) HIPX(dalloc(&Vt_, elemsVt_));
	} else {
		HIPX(dalloc(&t_vwh_, mpad_));
		HIPX(dalloc(&t_kl_, mpad_));
		HIPX(dalloc(&rowsum_part_, (std::max(mpad_, npad_) / 128) * RP_));
		HIPX(dalloc(&sW_, RP_));
		HIPX(dalloc(&sH_, RP_));
		HIPX(dalloc(&kl_scale_, RP_));
		HIPX(hipHostMalloc((void**)&pin_kl_, sizeof(T) * (2 * (size_t)m_ + 3 * (size_t)RP_)));
	}
	HIPX(dalloc(&Wt_, panelW));
	HIPX(dalloc(&H_, panelH));
	// wide fp32 panels: scratch for the split image of the r x r matrix the update kernel multiplies with (kernels_wide.hip)
	if (std::is_same<T, float>::value && panel_update_wide_available(RP_) && std::getenv("NMFAMD_FORCE_VALU") == nullptr &&
	    tuning_env("NMFAMD_WIDE_FP32_MFMA") == nullptr)
	{
		const size_t qb = 3 * 16 * (size_t)(RP_ / 16 + 1) * (RP_ / 32) * 64;
		HIPX(hipMalloc(&qx3_, qb));
		HIPX(hipMemsetAsync(qx3_, 0, qb, stream_));      // (the closing all-zero K-step: k_smooth_gram writes the others only)
	}
	HIPX(dalloc(&slabs_, slab_elems));
	HIPX(dalloc(&numW_, panelW));
	HIPX(dalloc(&G_, rr));
	HIPX(dalloc(&G2_, rr));
	HIPX(dalloc(&HHt_, rr));
	HIPX(dalloc(&Qinv_, rr));
	HIPX(dalloc(&gram_part_, rr * gram_parts_));
	HIPX(dalloc(&sumsq_part_, ((long)std::max(panel_update_parts(RP_, sizeof(T), (int)mpad_), (int)(mpad_ / panel_update_rows(RP_, sizeof(T)))) + 16) * RP_));
	// the two error-term vectors share one allocation (and one pinned landing buffer): ONE device-to-host copy per error iteration
	ps_stride_ = std::max<long>(npad_, RP_);
	HIPX(dalloc(&psN_, ps_stride_ + RP_));
	psR_ = psN_ + ps_stride_;
	HIPX(dalloc(&stage_, std::max(mpad_, npad_) * RP_));
	HIPX(hipMalloc((void**)&range_flag_, sizeof(int)));
	if constexpr (std::is_same<T, float>::value) {
		tri_ = bf16_ && tri_kernels_available(RP_) && (alg_ == ALG_MU || alg_ == ALG_NSNMF);
		if (tri_) {
			HIPX(hipMalloc((void**)&gram_tri_part_, sizeof(float) * (size_t)std::max<long>(gram_tri_partial_elems(num_cus_), (long)(TRI_PASSENGERS / 2) * 36 * 1024)));
			HIPX(hipMalloc((void**)&tri_ride_counters_, 64));
			HIPX(hipMemsetAsync(tri_ride_counters_, 0, 64, stream_));
			HIPX(dalloc(&Gw_raw_, rr));
			HIPX(dalloc(&Gh_raw_, rr));
			HIPX(dalloc(&colsq_, (long)RP_ * colsq_stage_parts()));
			tri_w_den_bf16_ = tuning_env("NMFAMD_TRI_FP32_DEN") == nullptr;
		}
	}
	if (alg_ == ALG_NSNMF) {
		HIPX(dalloc(&Ws_, panelW));                       // (tri_: only get_factors() materialises W S)
		if (!tri_) HIPX(dalloc(&Hs_, panelH));
	}
	if (alg_ >= ALG_GDCLS && alg_ <= ALG_AHCLS) {
		HIPX(dalloc(&Wold_, panelW));
		HIPX(dalloc(&rowdot_part_, (long)ROW_DOT_GROUPS * RP_));
		HIPX(hipMalloc((void**)&inv_work_, sizeof(double) * 2 * (size_t)r_ * r_));
		if (tuning_env("NMFAMD_NO_OVERLAP") == nullptr) {
			HIPX(hipStreamCreateWithFlags(&aux_, hipStreamNonBlocking));
			HIPX(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
			HIPX(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
			overlap_inverse_ = true;
		}
	}
	gram_image_ = x3_ && fused_capable() && RP_ == 64 && std::getenv("NMFAMD_GRAM_PARTIALS") == nullptr;
	one_pass_ = one_pass_ && x3_ && gram_image_ && one_image_ && img_th_ == 16;
#ifdef NMFAMD_DIAG_BUILD
	if (one_pass_) {
		HIPX(hipMalloc(&op_part_, onepass_part_bytes()));
		HIPX(hipMalloc(&op_hfrag_, onepass_hfrag_bytes()));
		HIPX(hipMalloc((void**)&op_ctl_, 64));
		HIPX(hipMalloc((void**)&op_slabs_, sizeof(float) * (size_t)ONEPASS_XCDS * RP_ * mpad_));
		HIPX(hipMalloc((void**)&op_hh_part_, sizeof(float) * 4096 * (size_t)(ONEPASS_XCDS * ONEPASS_GROUP)));
		HIPX(hipMalloc((void**)&op_ps4_, sizeof(float) * 4 * (size_t)npad_));
		HIPX(dalloc(&op_H2_, panelH));
		HIPX(hipMemsetAsync(op_ps4_, 0, sizeof(float) * 4 * (size_t)npad_, stream_));
		HIPX(hipMemsetAsync(op_part_, 0, onepass_part_bytes(), stream_));
		HIPX(hipMemsetAsync(op_hfrag_, 0, onepass_hfrag_bytes(), stream_));
		HIPX(hipMemsetAsync(op_ctl_, 0, 64, stream_));
		HIPX(hipMemsetAsync(op_slabs_, 0, sizeof(float) * (size_t)ONEPASS_XCDS * RP_ * mpad_, stream_));
		HIPX(hipHostMalloc((void**)&pin_abort_, sizeof(unsigned)));
		*pin_abort_ = 0;
		op_seq_ = 0;
		if (tuning_env("NMFAMD_ONEPASS_STAMPS") != nullptr) {
			HIPX(hipMalloc((void**)&op_stamps_, sizeof(unsigned long long) * 16 * 8 * ONEPASS_XCDS * ONEPASS_GROUP));
			HIPX(hipMemsetAsync(op_stamps_, 0, sizeof(unsigned long long) * 16 * 8 * ONEPASS_XCDS * ONEPASS_GROUP, stream_));
		}
	}
#endif
	if (fused_capable() || gram_from_update()) {
		HIPX(hipMalloc((void**)&gramW_part_, sizeof(float) * 4096 * (size_t)(mpad_ / 64)));
		HIPX(hipMalloc((void**)&Graw64_, sizeof(float) * 4096));      // the reduced, unscaled W^T W of normalize_w's one-launch form
		HIPX(hipMalloc((void**)&gramH_part_, sizeof(float) * 4096 * (size_t)(npad_ / 64)));
		HIPX(hipMalloc((void**)&scale_, sizeof(float) * 64));
		HIPX(hipMalloc((void**)&Gpart_, sizeof(float) * 4096 * GRAM_KSPLIT_MAX));
		HIPX(hipMemsetAsync(Gpart_, 0, sizeof(float) * 4096 * GRAM_KSPLIT_MAX, stream_));      // (spread form: a tile's absent pieces are never written)
		// W^T W of a whole problem (one K slice): the ten tiles' K ranges dealt to all sixteen passengers (gram_image.h) -- with a tile each, ten of them were the
		// last workgroups of config 2's W^T V launch (37 - 40 us against 33 - 36.5 for the product blocks, profiles/r05_update_tail.md).  NMFAMD_GRAM_SPREAD=0
		// (measurement builds) restores that form.
		gram_spread_ = gram_ksplit_ == 1 && !(tuning_env("NMFAMD_GRAM_SPREAD") != nullptr && std::atoi(tuning_env("NMFAMD_GRAM_SPREAD")) == 0);
		HIPX(hipMalloc((void**)&gram_spread_counter_, 64));
		HIPX(hipMemsetAsync(gram_spread_counter_, 0, 64, stream_));
		HIPX(hipMalloc((void**)&wsq_part_, sizeof(float) * 64 * (size_t)(mpad_ / 32)));
		HIPX(hipMemsetAsync(wsq_part_, 0, sizeof(float) * 64 * (size_t)(mpad_ / 32), stream_));
	}
	if (fused32w_capable()) {
		HIPX(hipMalloc((void**)&f32w_scale_, sizeof(float) * (size_t)RP_));
		HIPX(hipMemsetAsync(f32w_scale_, 0, sizeof(float) * (size_t)RP_, stream_));
		// The Gram slices as passengers of the product launch (gram_wide.h) where the product's first round of workgroups leaves CUs free for them and their chain --
		// a slice's K-steps at ~0.75 us each under the product's memory stream (measured: 625 steps in one slice made a 330 us launch 460) -- stays inside the launch,
		// or is shorter than the launch of their own it replaces (~10 us); otherwise k_gram_wide_x3.  Measured (profiles/r06_f32w_ride_ab.txt), 10 000 x 5 000:
		// r = 128 184.4 -> 166.7 us per iteration, r = 158 336.3 -> 309.7; 4096 x 165, r = 158: 103.7 -> 83.6; forced at r = 500 (one slice per super-block): 709 -> 967.
		// NMFAMD_F32W_RIDE = 0 / 1 (measurement builds): never / wherever a CU is free.
		const int nbk = RP_ / 128, nsuper = nbk * (nbk + 1) / 2;
		const char* force = tuning_env("NMFAMD_F32W_RIDE");
		auto ride = [&](const FactorProductPlan& p, long len) {
			const int wgs = p.xtiles * p.splits;      // (per layer of 128 columns; the passengers sit in the first layer)
			if (wgs >= num_cus_ || (force != nullptr && std::atoi(force) == 0)) return 0;
			const int slices = std::min((num_cus_ - wgs) / nsuper, gram_wide_fused_parts(RP_, (int)len, gram_parts_));
			if (slices < 1) return 0;
			const long steps = ((len + 15) / 16 + slices - 1) / slices;
			const double chain_us = 5.0 + 0.75 * (double)steps;
			const double product_us = 6.0 + (double)nbk * sizeof(T) * (double)mpad_ * (double)npad_ / 3.2e6;      // (65 us per layer of 128 columns at config 2's shape)
			return (chain_us <= 0.8 * product_us || steps <= 8 || (force != nullptr && std::atoi(force) == 1)) ? slices : 0;
		};
		f32w_ride_h_ = ride(planHx_, m_);
		f32w_ride_w_ = ride(planWx_, n_);
	}
	if (fused64_capable()) {
		// Gram passengers of the two product launches (kernels_f64.hip, gram_ride_f64): as many K slices per 64 x 64 super-block as the product's grid leaves CUs for,
		// at most 16 (the level-1 finisher adds them one after the other), at least 4 (a grid that fills the chip: the passengers queue behind the product blocks)
		const int nbk = RP_ / 64, nsuper = nbk * (nbk + 1) / 2;
		auto slices = [&](const FactorProductPlan& p, long len) {
			const int room = (num_cus_ - (p.half_tiles ? 2 : 1) * p.xtiles * p.splits * p.chunks - nbk) / nsuper;
			const int by_len = (int)std::max<long>(1, (len + 3) / 4 / 16);       // (at least 8 K-steps of four rows per wave half)
			return std::max(1, std::min(std::min(16, by_len), std::max(4, room)));
		};
		f64_slices_h_ = slices(planH_, m_);      // W^T W beside W^T V
		f64_slices_w_ = slices(planW_, n_);      // H H^T beside V H^T
		HIPX(hipMalloc((void**)&f64_scale_, sizeof(double) * (size_t)RP_));
		HIPX(hipMalloc((void**)&f64_partial_, sizeof(double) * 4096 * (size_t)nsuper * (size_t)std::max(f64_slices_h_, f64_slices_w_)));
		HIPX(hipMalloc((void**)&f64_counters_, sizeof(unsigned) * (size_t)(nsuper + 1)));
		HIPX(hipMemsetAsync(f64_counters_, 0, sizeof(unsigned) * (size_t)(nsuper + 1), stream_));
		if (tuning_env("NMFAMD_RIDE64_PID_ORDER") == nullptr) {
			std::vector<int> items;
			gram_ride_f64_items(planH_, RP_, f64_slices_h_, items);
			HIPX(hipMalloc((void**)&f64_items_h_, sizeof(int) * items.size()));
			HIPX(hipMemcpy(f64_items_h_, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
			gram_ride_f64_items(planW_, RP_, f64_slices_w_, items);
			HIPX(hipMalloc((void**)&f64_items_w_, sizeof(int) * items.size()));
			HIPX(hipMemcpy(f64_items_w_, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
		}
		if (tuning_env("NMFAMD_F64_STAMPS") != nullptr) {
			HIPX(hipMalloc((void**)&f64_stamps_, sizeof(unsigned long long) * 4 * 4096 * 8));
			HIPX(hipMemsetAsync(f64_stamps_, 0, sizeof(unsigned long long) * 4 * 4096 * 8, stream_));
		}
	}
	HIPX(hipHostMalloc((void**)&pin_psN_, sizeof(T) * (size_t)(ps_stride_ + RP_)));
	pin_psR_ = pin_psN_ + ps_stride_;
	{
		void* dp = nullptr;       // the device's view of the pinned buffer (fetch_error_terms writes it from a kernel)
		if (hipHostGetDevicePointer(&dp, pin_psN_, 0) == hipSuccess) pin_psN_dev_ = static_cast<T*>(dp); else { (void)hipGetLastError(); pin_psN_dev_ = nullptr; }
	}
	HIPX(hipEventCreateWithFlags(&err_event_, hipEventDisableTiming));
	HIPX(hipStreamSynchronize(stream_));
	return ST_OK;
}

// ---- upload / download --------------------------------------------------------------------

// Column-major staging image of V (ld = mpad_) -> the resident images and the tr(V^T V) terms.
// MFMA path (fp32): V_ and Vt_ are x-TILED (kernels.hip, k_factor_product_f32); the staging image
// is a temporary.  Generic path: V_ IS the column-major image and Vt_ its plain transpose.
template <typename T>
Status Engine<T>::finish_upload(T* Vcol) {
	int odd_values = 0;
	h_product_ahead_ = false; f64_product_ahead_ = false;
	HIPX(hipMemsetAsync(range_flag_, 0, sizeof(int), stream_));
	HIPX(launch_column_sumsq<T>(Vcol, mpad_, m_, n_, psN_, stream_, x3_ ? range_flag_ : nullptr));
	h_vtv_.resize(n_);
	HIPX(hipMemcpyAsync(h_vtv_.data(), psN_, sizeof(T) * n_, hipMemcpyDeviceToHost, stream_));
	HIPX(hipMemcpyAsync(&odd_values, range_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_));
	if (bf16_) {
		if constexpr (std::is_same<T, float>::value) {
			HIPX(launch_pack_stream_bf16(Vcol, mpad_, m_, n_, false, Vb_, planW_.xtiles, ksW_, stream_));
			HIPX(launch_pack_stream_bf16(Vcol, mpad_, n_, m_, true, Vtb_, planH_.xtiles, ksH_, stream_));
		}
	} else if (tiled_) {
		HIPX(hipMemsetAsync(V_, 0, sizeof(T) * (size_t)elemsV_, stream_));
		HIPX(launch_tile<T>(Vcol, mpad_, m_, n_, V_, strideV_, img_th_, false, stream_));
		if (!one_image_) {
			HIPX(hipMemsetAsync(Vt_, 0, sizeof(T) * (size_t)elemsVt_, stream_));
			HIPX(launch_tile_transposed<T>(Vcol, mpad_, m_, n_, Vt_, strideVt_, planH_.th, stream_));
		}
	} else {
		HIPX(launch_transpose<T>(Vcol, mpad_, m_, n_, Vt_, npad_, stream_));
	}
	HIPX(hipStreamSynchronize(stream_));
	std::sort(h_vtv_.begin(), h_vtv_.end());
	if (x3_ && odd_values != 0) {
		last_error_ = "V holds values outside the exact range of the split-operand product (not finite, |v| > 2^126 or 0 < |v| < 2^-100): use precision = -1";
		return ST_VALUE_RANGE;
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::upload_dense(const T* V, long ld) {
	if (!V || ld < m_) return ST_INVALID;
	if (sparse_) {
		// dense input on the sparse path: the non-zero entries become the stored entries
		std::vector<int> rows, cols; std::vector<T> vals;
		for (int j = 0; j < n_; ++j)
			for (int i = 0; i < m_; ++i) {
				const T v = V[(size_t)j * ld + i];
				if (v != T(0)) { rows.push_back(i); cols.push_back(j); vals.push_back(v); }
			}
		return upload_triplets(rows, cols, vals);
	}
	T* Vcol = V_;
	const bool staged = tiled_ || bf16_;
	if (staged) {
		HIPX(hipMalloc((void**)&Vcol, sizeof(T) * (size_t)(mpad_ * npad_)));
		hipError_t e = hipMemsetAsync(Vcol, 0, sizeof(T) * (size_t)(mpad_ * npad_), stream_);
		if (e != hipSuccess) { (void)hipFree(Vcol); return hip_fail(e, "hipMemsetAsync(staging)"); }
	}
	hipError_t e = hipMemcpy2DAsync(Vcol, mpad_ * sizeof(T), V, ld * sizeof(T), m_ * sizeof(T), n_, hipMemcpyHostToDevice, stream_);
	Status st = e == hipSuccess ? finish_upload(Vcol) : hip_fail(e, "hipMemcpy2DAsync(V)");
	if (staged) (void)hipFree(Vcol);
	return st;
}

template <typename T>
Status Engine<T>::upload_sparse(int format, const T* values, const int* a, const int* b, long nnz, int base) {
	if (format < 1 || format > 3 || nnz < 0 || (nnz > 0 && (!values || !a || !b))) return ST_INVALID;
	if (sparse_) {
		// the images are built on the device (kernels_sparse_setup.hip); what that path does not cover -- entries outside the matrix (dropped below), pointer
		// arrays that do not ascend, a row or column longer than its LDS sort takes, no entries at all -- and NMFAMD_SPARSE_SETUP=host take the host path
		const char* where = std::getenv("NMFAMD_SPARSE_SETUP");
		if (!(where != nullptr && std::strcmp(where, "host") == 0) && nnz > 0) {
			bool fallback = false;
			const Status st = upload_sparse_device(format, values, a, b, nnz, base, &fallback);
			if (!fallback) return st;
		}
		sparse_setup_on_device_ = false;
		// expand to 0-based (row, column, value) triplets with the caller's index base applied exactly
		// (reference: cusparseSetMatIndexBase, Matrix.h:158-160,184-186,215-217); out-of-range entries are dropped
		std::vector<int> rows, cols; std::vector<T> vals;
		rows.reserve(nnz); cols.reserve(nnz); vals.reserve(nnz);
		auto push = [&](long i, long j, T v) { if (i >= 0 && i < m_ && j >= 0 && j < n_) { rows.push_back((int)i); cols.push_back((int)j); vals.push_back(v); } };
		if (format == 1) { for (int i = 0; i < m_; ++i) for (long p = (long)a[i] - base; p < (long)a[i + 1] - base && p < nnz; ++p) if (p >= 0) push(i, (long)b[p] - base, values[p]); }
		else if (format == 2) { for (int j = 0; j < n_; ++j) for (long p = (long)a[j] - base; p < (long)a[j + 1] - base && p < nnz; ++p) if (p >= 0) push((long)b[p] - base, j, values[p]); }
		else { for (long p = 0; p < nnz; ++p) push((long)a[p] - base, (long)b[p] - base, values[p]); }
		return upload_triplets(rows, cols, vals);
	}
	T* d_val = nullptr; int *d_a = nullptr, *d_b = nullptr;
	T* Vcol = nullptr;
	const int outer = format == 1 ? m_ : n_;
	const long na = format == 3 ? nnz : (long)outer + 1;
	Status st = ST_OK;
	do {
		hipError_t e;
		if ((e = hipMalloc((void**)&d_val, sizeof(T) * (size_t)std::max<long>(nnz, 1))) != hipSuccess ||
		    (e = hipMalloc((void**)&d_a, sizeof(int) * (size_t)std::max<long>(na, 1))) != hipSuccess ||
		    (e = hipMalloc((void**)&d_b, sizeof(int) * (size_t)std::max<long>(nnz, 1))) != hipSuccess) { st = hip_fail(e, "hipMalloc(sparse staging)"); break; }
		if (nnz > 0) {
			if ((e = hipMemcpyAsync(d_val, values, sizeof(T) * nnz, hipMemcpyHostToDevice, stream_)) != hipSuccess ||
			    (e = hipMemcpyAsync(d_b, b, sizeof(int) * nnz, hipMemcpyHostToDevice, stream_)) != hipSuccess) { st = hip_fail(e, "hipMemcpyAsync(sparse)"); break; }
		}
		if (na > 0 && (e = hipMemcpyAsync(d_a, a, sizeof(int) * na, hipMemcpyHostToDevice, stream_)) != hipSuccess) { st = hip_fail(e, "hipMemcpyAsync(sparse ptr)"); break; }
		if ((tiled_ || bf16_) && (e = hipMalloc((void**)&Vcol, sizeof(T) * (size_t)(mpad_ * npad_))) != hipSuccess) { Vcol = nullptr; st = hip_fail(e, "hipMalloc(staging)"); break; }
		if (!(tiled_ || bf16_)) Vcol = V_;
		if ((e = hipMemsetAsync(Vcol, 0, sizeof(T) * (size_t)(mpad_ * npad_), stream_)) != hipSuccess) { st = hip_fail(e, "hipMemsetAsync(V)"); break; }
		// CSR: ptr = a (rowPtr), idx = b (columns); CSC: ptr = a (columnPtr), idx = b (rows); COO: idx = a (rows), idx2 = b (columns)
		if (format == 3) e = launch_densify<T>(3, d_val, nullptr, d_a, d_b, nnz, 0, base, Vcol, mpad_, m_, n_, stream_);
		else e = launch_densify<T>(format, d_val, d_a, d_b, nullptr, nnz, outer, base, Vcol, mpad_, m_, n_, stream_);
		if (e != hipSuccess) { st = hip_fail(e, "densify"); break; }
		st = finish_upload(Vcol);
	} while (0);
	if ((tiled_ || bf16_) && Vcol) (void)hipFree(Vcol);
	if (d_val) (void)hipFree(d_val);
	if (d_a) (void)hipFree(d_a);
	if (d_b) (void)hipFree(d_b);
	return st;
}

// The CSR and CSC images built on the device from the caller's arrays as they are (kernels_sparse_setup.hip; reference: the device-side conversions of
// source/common/Matrix.h:145-232).  *fallback: the input needs the host path (nothing of this engine's state has been touched then, except freed images).
template <typename T>
Status Engine<T>::upload_sparse_device(int format, const T* values, const int* a, const int* b, long nnz, int base, bool* fallback) {
	*fallback = false;
	if (nnz >= (1l << 31)) return ST_INVALID;
	const int outer = format == 1 ? m_ : n_;
	const long na = format == 3 ? nnz : (long)outer + 1;
	const int segs = std::max(m_, n_);
	// scratch: freed on every way out
	T* d_val = nullptr;
	int *d_a = nullptr, *d_b = nullptr, *row = nullptr, *col = nullptr, *items = nullptr, *rowq = nullptr, *counts = nullptr, *small = nullptr;
	T* d_vtv = nullptr;
	double* d_colsum = nullptr;
	std::vector<void*> owned;
	auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16)); if (e == hipSuccess) owned.push_back(*p); else *p = nullptr; return e; };
	auto disown = [&](void* p) { for (auto& q : owned) if (q == p) q = nullptr; };
	struct Cleanup { std::vector<void*>& v; ~Cleanup() { for (void* p : v) if (p) (void)hipFree(p); } } cleanup{owned};
	auto need_host = [&]() { *fallback = true; return ST_OK; };

	void** old[] = {(void**)&csr_ptr_, (void**)&csr_idx_, (void**)&csc_ptr_, (void**)&csc_idx_, (void**)&csc_from_csr_, (void**)&csr_val_, (void**)&csc_val_, (void**)&q_, (void**)&q2_};
	for (void** o : old) { if (*o) (void)hipFree(*o); *o = nullptr; }
	nnz_ = 0;

	const size_t ni = sizeof(int) * (size_t)nnz, nv = sizeof(T) * (size_t)nnz;
	HIPX(alloc((void**)&d_val, nv)); HIPX(alloc((void**)&d_a, sizeof(int) * (size_t)na)); HIPX(alloc((void**)&d_b, ni));
	HIPX(alloc((void**)&row, ni)); HIPX(alloc((void**)&col, ni));
	HIPX(alloc((void**)&counts, sizeof(int) * (size_t)segs)); HIPX(alloc((void**)&small, sizeof(int) * 4));
	HIPX(hipMemcpyAsync(d_val, values, nv, hipMemcpyHostToDevice, stream_));
	HIPX(hipMemcpyAsync(d_a, a, sizeof(int) * (size_t)na, hipMemcpyHostToDevice, stream_));
	HIPX(hipMemcpyAsync(d_b, b, ni, hipMemcpyHostToDevice, stream_));
	HIPX(hipMemsetAsync(small, 0, sizeof(int) * 4, stream_));
	int* flags = small;                       // [0] flags, [1] longest row, [2] longest column
	HIPX(launch_sp_expand(format, d_a, d_b, nnz, outer, base, m_, n_, row, col, flags, stream_));
	HIPX(hipMalloc((void**)&csr_ptr_, sizeof(int) * (size_t)(m_ + 1)));
	HIPX(hipMalloc((void**)&csc_ptr_, sizeof(int) * (size_t)(n_ + 1)));
	HIPX(launch_sp_histogram_scan(row, nnz, m_, counts, csr_ptr_, small + 1, stream_));
	int h_small[4] = {0, 0, 0, 0};
	HIPX(hipMemcpyAsync(h_small, small, sizeof(int) * 4, hipMemcpyDeviceToHost, stream_));
	HIPX(hipStreamSynchronize(stream_));
	if ((h_small[0] & 3) != 0) return need_host();
	(void)hipFree(d_a); disown(d_a); d_a = nullptr;
	(void)hipFree(d_b); disown(d_b); d_b = nullptr;

	HIPX(alloc((void**)&items, ni));
	if ((h_small[0] & 4) != 0) {
		// not in (row, column) order (COO, CSC, an unsorted CSR): the entries' positions by row, every row by (column, position)
		if (h_small[1] > sp_segment_sort_capacity()) return need_host();
		HIPX(alloc((void**)&rowq, ni));
		HIPX(hipMalloc((void**)&csr_idx_, ni)); HIPX(hipMalloc((void**)&csr_val_, nv));
		HIPX(launch_sp_scatter_sort(row, nnz, m_, csr_ptr_, counts, items, col, h_small[1], stream_));
		HIPX(launch_sp_gather_csr<T>(items, row, col, d_val, nnz, csr_idx_, csr_val_, rowq, stream_));
	} else {
		// the input IS the CSR image (0-based by now): its buffers are adopted as they are
		csr_idx_ = col; disown(col);
		csr_val_ = d_val; disown(d_val); d_val = nullptr;
		rowq = row;
	}
	HIPX(launch_sp_histogram_scan(csr_idx_, nnz, n_, counts, csc_ptr_, small + 2, stream_));
	HIPX(hipMemcpyAsync(h_small, small, sizeof(int) * 4, hipMemcpyDeviceToHost, stream_));
	HIPX(hipStreamSynchronize(stream_));
	if (h_small[2] > sp_segment_sort_capacity()) return need_host();
	// the CSR positions by column, every column's list ascending = ascending row, duplicates in CSR order
	HIPX(launch_sp_scatter_sort(csr_idx_, nnz, n_, csc_ptr_, counts, items, nullptr, h_small[2], stream_));
	HIPX(hipMalloc((void**)&csc_idx_, ni)); HIPX(hipMalloc((void**)&csc_val_, nv));
	HIPX(launch_sp_gather_csc<T>(items, rowq, csr_val_, nnz, csc_idx_, csc_val_, stream_));
	const bool two_pass = tuning_env("NMFAMD_KL_TWO_PASS") != nullptr;      // (round 1's two-pass KL step, a measurement switch: quotient buffers + the CSR -> CSC permutation)
	if (two_pass) { csc_from_csr_ = items; disown(items); HIPX(hipMalloc((void**)&q_, nv)); HIPX(hipMalloc((void**)&q2_, nv)); }
	// tr(V^T V) terms per column (accumulated in T like the trace kernel, in the image's order) and sum(V) for the KL divergence
	HIPX(alloc((void**)&d_vtv, sizeof(T) * (size_t)n_)); HIPX(alloc((void**)&d_colsum, sizeof(double) * (size_t)n_));
	HIPX(launch_sp_col_sumsq<T>(csc_ptr_, csc_val_, n_, d_vtv, d_colsum, stream_));
	h_vtv_.assign(n_, T(0));
	std::vector<double> colsum((size_t)n_);
	HIPX(hipMemcpyAsync(h_vtv_.data(), d_vtv, sizeof(T) * (size_t)n_, hipMemcpyDeviceToHost, stream_));
	HIPX(hipMemcpyAsync(colsum.data(), d_colsum, sizeof(double) * (size_t)n_, hipMemcpyDeviceToHost, stream_));
	HIPX(hipStreamSynchronize(stream_));
	std::sort(h_vtv_.begin(), h_vtv_.end());
	sum_v_ = 0;
	for (int j = 0; j < n_; ++j) sum_v_ += colsum[j];
	nnz_ = nnz;
	sparse_setup_on_device_ = true;
	if (Status st = setup_kl_blocks()) return st;
	HIPX(hipStreamSynchronize(stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::set_factors(const T* W, long ldw, const T* H, long ldh) {
	if (W) {
		if (ldw < m_) return ST_INVALID;
		fused_ready_ = false; w_pending_ = false; f32w_pending_ = false; f64_pending_ = false; f64_product_ahead_ = false; kl_scale_pending_ = false; gram_w_ready_ = false; wx3_valid_ = false; hx3_valid_ = false; wtb_valid_ = false; tri_gw_ready_ = false; qx3_holds_g_ = false; h_product_ahead_ = false;
		tri_scale_pending_ = false; tri_scale_from_gram_ = false; kl_sw_ready_ = false; w_rows_stale_ = false;
		// host m x r (column-major) -> staging m x r (ld mpad) -> Wt panel (element (c, i) at [i * RP + c])
		HIPX(hipMemcpy2DAsync(stage_, mpad_ * sizeof(T), W, ldw * sizeof(T), m_ * sizeof(T), r_, hipMemcpyHostToDevice, stream_));
		HIPX(hipMemsetAsync(Wt_, 0, sizeof(T) * (size_t)RP_ * mpad_, stream_));
		HIPX(launch_transpose<T>(stage_, mpad_, m_, r_, Wt_, RP_, stream_));
	}
	if (H) {
		hx3_valid_ = false; hb_valid_ = false;
		if (ldh < r_) return ST_INVALID;
		gram_h_partials_ = false;
		HIPX(hipMemsetAsync(H_, 0, sizeof(T) * (size_t)RP_ * npad_, stream_));
		HIPX(hipMemcpy2DAsync(H_, RP_ * sizeof(T), H, ldh * sizeof(T), r_ * sizeof(T), n_, hipMemcpyHostToDevice, stream_));
	}
	HIPX(hipStreamSynchronize(stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::get_factors(T* W, long ldw, T* H, long ldh) {
	if (one_pass_) { if (Status s = onepass_check()) return s; }
	if (Status s = materialize_w()) return s;
	if (W) {
		if (ldw < m_) return ST_INVALID;
		const T* src = Wt_;
		if (alg_ == ALG_NSNMF) {
			// storeFactorization returns W S (AlgorithmNonSmoothNMF.h:221-225)
			const T off = (T)prm_.theta / (T)(unsigned)r_;
			const T diag = (T)((1.0 - (T)prm_.theta) + off);
			HIPX(launch_smooth_panel<T>(Wt_, Ws_, RP_, r_, mpad_, off, diag, stream_));
			src = Ws_;
		}
		HIPX(launch_transpose<T>(src, RP_, r_, m_, stage_, mpad_, stream_));
		HIPX(hipMemcpy2DAsync(W, ldw * sizeof(T), stage_, mpad_ * sizeof(T), m_ * sizeof(T), r_, hipMemcpyDeviceToHost, stream_));
	}
	if (H) {
		if (ldh < r_) return ST_INVALID;
		HIPX(hipMemcpy2DAsync(H, ldh * sizeof(T), H_, RP_ * sizeof(T), r_ * sizeof(T), n_, hipMemcpyDeviceToHost, stream_));
	}
	HIPX(hipStreamSynchronize(stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::randomize_factors(unsigned seed, bool w, bool h, long h_first_column) {
	// The reference seeds W's and H's generators identically (RandomValueStrategy.cpp:53-69).
	if (w) { h_product_ahead_ = false; f64_product_ahead_ = false; kl_sw_ready_ = false; fused_ready_ = false; w_pending_ = false; f32w_pending_ = false; f64_pending_ = false; kl_scale_pending_ = false; gram_w_ready_ = false; wx3_valid_ = false; hx3_valid_ = false; wtb_valid_ = false; tri_gw_ready_ = false; qx3_holds_g_ = false; tri_scale_pending_ = false; tri_scale_from_gram_ = false; w_rows_stale_ = false; }
	if (h) { gram_h_partials_ = false; hx3_valid_ = false; hb_valid_ = false; }
	if (w) HIPX(launch_fill_uniform<T>(Wt_, RP_, r_, m_, mpad_, seed, stream_));
	if (h) HIPX(launch_fill_uniform<T>(H_, RP_, r_, n_, npad_, seed, stream_, h_first_column));
	return ST_OK;
}

// ---- the two big products ------------------------------------------------------------------

template <typename T>
void Engine<T>::record_begin(int kind) {
	if (!timing_ || !timing_now_) return;
	if (ev_used_ + 2 > ev_.size()) {
		for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev_.push_back(e); }
	}
	if (ev_kind_.size() < ev_.size() / 2) ev_kind_.resize(ev_.size() / 2, 0);
	ev_kind_[ev_used_ / 2] = (char)kind;
	(void)hipEventRecord(ev_[ev_used_], stream_);
}

template <typename T>
bool Engine<T>::timed_launch_events(int kind, hipEvent_t* start, hipEvent_t* stop) {
	*start = *stop = nullptr;
	if (!timing_ || !timing_now_) return false;
	if (ev_used_ + 2 > ev_.size()) {
		for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return false; ev_.push_back(e); }
	}
	if (ev_kind_.size() < ev_.size() / 2) ev_kind_.resize(ev_.size() / 2, 0);
	ev_kind_[ev_used_ / 2] = (char)kind;
	*start = ev_[ev_used_]; *stop = ev_[ev_used_ + 1];
	return true;
}

template <typename T>
void Engine<T>::record_end() {
	if (!timing_ || !timing_now_ || ev_used_ + 2 > ev_.size()) return;
	(void)hipEventRecord(ev_[ev_used_ + 1], stream_);
	ev_used_ += 2;
}

template <typename T>
void Engine<T>::dominant_stats(double* total_ms, long* launches, double* pair_overhead_ms, double* kind_ms, long* kind_launches) {
	double tot = 0; long cnt = 0;
	double km[2] = {0, 0}; long kc[2] = {0, 0};
	(void)hipStreamSynchronize(stream_);
	for (size_t i = 0; i + 1 < ev_used_; i += 2) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, ev_[i], ev_[i + 1]) == hipSuccess) {
			tot += ms; ++cnt;
			const int k = (i / 2 < ev_kind_.size() && ev_kind_[i / 2] != 0) ? 1 : 0;
			km[k] += ms; ++kc[k];
		}
	}
	ev_used_ = 0;
	if (total_ms) *total_ms = tot;
	if (launches) *launches = cnt;
	if (kind_ms) { kind_ms[0] = km[0]; kind_ms[1] = km[1]; }
	if (kind_launches) { kind_launches[0] = kc[0]; kind_launches[1] = kc[1]; }
	if (pair_overhead_ms) {
		// what an event pair reports with NOTHING between the two records (idle stream, smallest of eight):
		// the part of every sample above that is not kernel time
		double best = -1;
		if (ev_.size() >= 2) {
			for (int i = 0; i < 8; ++i) {
				(void)hipEventRecord(ev_[0], stream_);
				(void)hipEventRecord(ev_[1], stream_);
				(void)hipEventSynchronize(ev_[1]);
				float ms = 0;
				if (hipEventElapsedTime(&ms, ev_[0], ev_[1]) == hipSuccess && (best < 0 || ms < best)) best = ms;
			}
		}
		*pair_overhead_ms = best < 0 ? 0.0 : best;
	}
}

// Passenger arguments for the Gram matrix of W (of_w) or H of the rank-64 fast path: from the split image of the panel
// (gram_image_) or from the partial matrices the 64-column update kernel left behind.
template <typename T>
GramReduceArgs Engine<T>::gram_args(bool of_w, float* G, float* scale, int normalize) const {
	GramReduceArgs rg = {of_w ? gramW_part_ : gramH_part_, (int)((of_w ? mpad_ : npad_) / 64), G, scale, normalize};
	if (gram_image_) {
		rg.partials = nullptr; rg.parts = 0; rg.image = of_w ? Wx3_ : Hx3_; rg.image_ks = of_w ? ksH_ : ksW_;
		// the pending column scale of W from the sums of squares its update left (one vector per 32 panel rows)
		if (of_w && normalize != 0 && wsq_part_ != nullptr) { rg.colsq_part = wsq_part_; rg.colsq_parts = (int)(mpad_ / 32); }
		// W^T W beside the W^T V launch of a column shard: K slices into Gpart_, added and scaled by the H update (mu64_update)
		if (of_w && (const void*)G == (const void*)G_ && gram_ksplit_ > 1 && Gpart_ != nullptr && (normalize == 0 || rg.colsq_part != nullptr)) { rg.ksplit = gram_ksplit_; rg.G = Gpart_; }
		// ... of a whole problem: the sixteen passengers share the ten tiles' K ranges evenly (gram_image.h, spread form), pieces into Gpart_ for the same consumer
		else if (of_w && (const void*)G == (const void*)G_ && gram_spread_ && Gpart_ != nullptr && (normalize == 0 || rg.colsq_part != nullptr)) { rg.spread = 1; rg.G = Gpart_; rg.spread_out = G; rg.spread_counter = gram_spread_counter_; }
	}
	return rg;
}

template <typename T>
Status Engine<T>::standalone_gram(const GramReduceArgs& rg) {
	if (rg.image != nullptr) HIPX(launch_gram_image_args(rg, stream_));
	else HIPX(launch_mu64_gram_reduce(rg, stream_));
	return ST_OK;
}

// U_H / U_W of the rank-64 fast path (kernels_mu64.hip)
template <typename T>
Status Engine<T>::mu64_update(bool is_w, const T* slabs, int S, long slab_stride, const T* Q, bool compute_error, const PeerSlabs* peers) {
	if constexpr (std::is_same<T, float>::value) {
		const float eps = std::numeric_limits<float>::epsilon();
		float* P = is_w ? Wt_ : H_;
		// (ps_direct_: the update kernels write their error terms straight into the pinned host buffer -- no copy launch behind them, iterate_mu64)
		float* ps = ps_direct_ ? (is_w ? pin_psN_dev_ + ps_stride_ : pin_psN_dev_) : (is_w ? psR_ : psN_);
		const int len = is_w ? m_ : n_, len_pad = (int)(is_w ? mpad_ : npad_);
		void* xo = x3_ ? (is_w ? Wx3_ : Hx3_) : nullptr;
		const int xks = is_w ? ksH_ : ksW_;
		if (peers != nullptr && !gram_image_) return ST_INVALID;
		// nsNMF: the smoothing matrix goes around the H update's r x r product (the launches on either side work on the unsmoothed W image and on the image of S H)
		SmoothAround ns;
		ns.off = (float)prm_.theta / (float)(unsigned)r_; ns.diag = (float)((1.0 - (float)prm_.theta) + ns.off); ns.r = r_;
		const bool smooth = alg_ == ALG_NSNMF && !is_w;
		if (alg_ == ALG_NSNMF && !gram_image_) return ST_INVALID;
		if (gram_image_) HIPX(launch_mu64_update32(is_w ? 1 : 0, P, slabs, S, slab_stride, Q, scale_, eps, ps, len, len_pad, is_w ? G_ : nullptr, compute_error ? 1 : 0, stream_, xo, xks, peers,
		                                           is_w ? wsq_part_ : nullptr, (!is_w && (const void*)Q == (const void*)Gpart_) ? gram_ksplit_ : 0, (!is_w && (const void*)Q == (const void*)Gpart_) ? G_ : nullptr,
		                                           smooth ? &ns : nullptr));
		else HIPX(launch_mu64_update(is_w ? 1 : 0, P, slabs, S, slab_stride, Q, scale_, eps, ps, len, len_pad, is_w ? gramW_part_ : gramH_part_, is_w ? G_ : nullptr,
		                             compute_error ? 1 : 0, stream_, xo, xks));
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::product_h(const T* F, const GramReduceArgs* rg, bool prepacked) {
	if (sparse_) {
		// W^T V as a row-gather SpMM over the CSC image: out(:, j) = sum_i V(i, j) F(:, i)
		if (rg) { if (Status st = standalone_gram(*rg)) return st; }
		record_begin();
		HIPX(launch_spmm_rows<T>(csc_ptr_, csc_idx_, csc_val_, F, RP_, slabs_, n_, (int)npad_, stream_));
		record_end();
		return ST_OK;
	}
	if constexpr (std::is_same<T, float>::value) {
		if (bf16_) {
			// bf16 operands: the factor panel is re-rounded and re-ordered for every product
			if (!prepacked) HIPX(launch_pack_panel_bf16(F, RP_, m_, Wtb_, ksH_, stream_));
			if (rg && rg->tri_frags == nullptr && planHb_.xtiles < GRAM_REDUCE_BLOCKS) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			record_begin();
#ifdef NMFAMD_DIAG_BUILD
			if (unsigned long long* st = bf_stamps(0)) set_factor_product_bf16_stamps(st);
#endif
			HIPX(launch_factor_product_bf16(planHb_, Vtb_, ksH_, Wtb_, RP_, slabs_, slab_stride_, stream_, rg));
			record_end();
			return ST_OK;
		}
		if (x3_) {
			if (!prepacked) HIPX(launch_pack_panel_x3(F, RP_, m_, Wx3_, ksH_, stream_));
			if (rg && rg->wide_P == nullptr && !passengers_ride(planHx_)) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			hipEvent_t e0, e1;
			const bool timed = timed_launch_events(0, &e0, &e1);
			if (one_image_) HIPX(launch_factor_product_x3(planHx_, V_, strideV_, Wx3_, RP_, slabs_, slab_stride_, stream_, rg, nullptr, true, img_th_, e0, e1));
			else HIPX(launch_factor_product_x3(planHx_, Vt_, strideVt_, Wx3_, RP_, slabs_, slab_stride_, stream_, rg, nullptr, false, 128, e0, e1));
			if (timed) timed_launch_done();
			return ST_OK;
		}
		if (tiled_) {
			if (rg && planH_.xtiles < GRAM_REDUCE_BLOCKS) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			record_begin();
			HIPX(launch_factor_product_f32(planH_, Vt_, strideVt_, F, RP_, slabs_, slab_stride_, stream_, rg));
			record_end();
			return ST_OK;
		}
	}
	if constexpr (std::is_same<T, double>::value) {
		if (tiled_) {
			record_begin();
			HIPX(launch_factor_product_f64(planH_, Vt_, strideVt_, F, RP_, slabs_, slab_stride_, stream_, ride64_));
			record_end();
			return ST_OK;
		}
	}
	record_begin();
	HIPX(launch_factor_product_valu<T>(Vt_, npad_, (int)npad_, m_, F, RP_, slabs_, stream_));
	record_end();
	return ST_OK;
}

template <typename T>
Status Engine<T>::product_w(const T* F, const GramReduceArgs* rg, T* single_slab_out, bool prepacked) {
	T* dest = (single_slab_out != nullptr && planW_.splits == 1) ? single_slab_out : slabs_;
	if (sparse_) {
		// (V H^T)^T over the CSR image: out(:, i) = sum_j V(i, j) F(:, j)
		if (rg) { if (Status st = standalone_gram(*rg)) return st; }
		record_begin(1);
		HIPX(launch_spmm_rows<T>(csr_ptr_, csr_idx_, csr_val_, F, RP_, dest, m_, (int)mpad_, stream_));
		record_end();
		return ST_OK;
	}
	if constexpr (std::is_same<T, float>::value) {
		if (bf16_) {
			if (!prepacked) HIPX(launch_pack_panel_bf16(F, RP_, n_, Hb_, ksW_, stream_));
			if (rg && rg->tri_frags == nullptr && planWb_.xtiles < GRAM_REDUCE_BLOCKS) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			record_begin(1);
#ifdef NMFAMD_DIAG_BUILD
			if (unsigned long long* st = bf_stamps(1)) set_factor_product_bf16_stamps(st);
#endif
			HIPX(launch_factor_product_bf16(planWb_, Vb_, ksW_, Hb_, RP_, dest, slab_stride_, stream_, rg));
			record_end();
			return ST_OK;
		}
		if (x3_) {
			if (!prepacked) HIPX(launch_pack_panel_x3(F, RP_, n_, Hx3_, ksW_, stream_));
			if (rg && rg->wide_P == nullptr && !passengers_ride(planWx_)) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			hipEvent_t e0, e1;
			const bool timed = timed_launch_events(1, &e0, &e1);
			HIPX(launch_factor_product_x3(planWx_, V_, strideV_, Hx3_, RP_, dest, slab_stride_, stream_, rg, nullptr, false, img_th_, e0, e1));
			if (timed) timed_launch_done();
			return ST_OK;
		}
		if (tiled_) {
			if (rg && planW_.xtiles < GRAM_REDUCE_BLOCKS) { if (Status st = standalone_gram(*rg)) return st; rg = nullptr; }
			record_begin(1);
			HIPX(launch_factor_product_f32(planW_, V_, strideV_, F, RP_, dest, slab_stride_, stream_, rg));
			record_end();
			return ST_OK;
		}
	}
	if constexpr (std::is_same<T, double>::value) {
		if (tiled_) {
			record_begin(1);
			HIPX(launch_factor_product_f64(planW_, V_, strideV_, F, RP_, dest, slab_stride_, stream_, ride64_));
			record_end();
			return ST_OK;
		}
	}
	record_begin(1);
	HIPX(launch_factor_product_valu<T>(V_, mpad_, (int)mpad_, n_, F, RP_, dest, stream_));
	record_end();
	return ST_OK;
}

template <typename T>
Status Engine<T>::normal_inverse(T* A, T offdiag, T diag) {
	HIPX(launch_inverse_small<T>(A, RP_, r_, Qinv_, inv_work_, offdiag, diag, stream_));
	return ST_OK;
}

// Fork: everything enqueued on the main stream so far (the Gram matrix, its saved copy) precedes the inverse;
// join: the main stream waits for the inverse before the update kernel reads Qinv_.  Between the two the main
// stream must not touch A, Qinv_ or inv_work_.
template <typename T>
Status Engine<T>::normal_inverse_fork(T* A, T offdiag, T diag) {
	if (!overlap_inverse_) return normal_inverse(A, offdiag, diag);
	HIPX(hipEventRecord(ev_fork_, stream_));
	HIPX(hipStreamWaitEvent(aux_, ev_fork_, 0));
	HIPX(launch_inverse_small<T>(A, RP_, r_, Qinv_, inv_work_, offdiag, diag, aux_));
	HIPX(hipEventRecord(ev_join_, aux_));
	return ST_OK;
}

// Split-operand product at padded rank 64: the sixteen passenger workgroups (Gram matrix from the split image, or the one that inverts) sit BEHIND the
// product blocks in the grid.  They ride when the product has at least sixteen x-tiles (config 2's shapes: the grid is several waves of workgroups and the
// passengers start as CUs drain) or when the whole grid fits the chip at once (column shards: few x-tiles, e.g. 5 x 26 + 16 workgroups at n = 625) --
// round 3 sent the latter to a stand-alone 16-workgroup launch of 28 us.
template <typename T>
bool Engine<T>::passengers_ride(const FactorProductPlan& plan) const {
	const int passengers = (&plan == &planHx_ && gram_ksplit_ > 1) ? GRAM_IMAGE_TILES * gram_ksplit_ : GRAM_REDUCE_BLOCKS;
	const int per_cu = plan.col_split == 2 ? 2 : 1;       // (128 x 32 workgroups: two to a CU)
	return RP_ == 64 && ((plan.xtiles >= GRAM_REDUCE_BLOCKS && passengers == GRAM_REDUCE_BLOCKS) || per_cu * plan.xtiles * plan.splits + passengers <= per_cu * num_cus_);
}

// fp32 product (native MFMA or split-operand kernel) at padded rank 64 with passenger blocks in its grid
template <typename T>
bool Engine<T>::inverse_rides(const FactorProductPlan& plan) const {
	return std::is_same<T, float>::value && tiled_ && !bf16_ && !sparse_ && RP_ == 64 && r_ <= 64 && plan.xtiles >= GRAM_REDUCE_BLOCKS &&
	       tuning_env("NMFAMD_NO_OVERLAP") == nullptr && tuning_env("NMFAMD_INVERSE_SIDE_STREAM") == nullptr;
}

template <typename T>
Status Engine<T>::normal_inverse_join() {
	if (!overlap_inverse_) return ST_OK;
	HIPX(hipStreamWaitEvent(stream_, ev_join_, 0));
	return ST_OK;
}

// ---- error terms ----------------------------------------------------------------------------

template <typename T>
Status Engine<T>::fetch_error_terms(int count_n) {
	finalize_error(false);   // the pinned buffers are about to be reused; an older fetch is long complete
	ps_last_direct_ = ps_direct_;
	if (error_terms_stay_) {
		// a sharded run gathers the terms of all ranks on the device (error_terms_to_device) and resolves them itself: no host copy, no event here
		err_count_ = count_n;
		return ST_OK;
	}
	// (a kernel that writes the pinned buffer, not hipMemcpyAsync: the runtime's copy idles the stream for ~18 us around its blit; NMFAMD_ERROR_MEMCPY=1 restores it)
	static const bool use_memcpy = tuning_env("NMFAMD_ERROR_MEMCPY") != nullptr;
	if (ps_direct_) { /* the update kernels of this iteration wrote the pinned buffer themselves */ }
	else if (use_memcpy || pin_psN_dev_ == nullptr) HIPX(hipMemcpyAsync(pin_psN_, psN_, sizeof(T) * (size_t)(ps_stride_ + r_), hipMemcpyDeviceToHost, stream_));
	else HIPX(launch_copy_small<T>(pin_psN_dev_, psN_, ps_stride_ + r_, stream_));
	HIPX(hipEventRecord(err_event_, stream_));
	err_pending_ = true;
	err_count_ = count_n;
	return ST_OK;
}

template <typename T>
long Engine<T>::error_terms_to_device(T* dst, long capacity) {
	const long cnt = (long)err_count_ + r_;
	if (err_count_ <= 0 || capacity < cnt) return -1;
	// (the last error iteration's update kernels wrote the pinned host buffer themselves, iterate_mu64: psN_ / psR_ on the device are an older iteration's -- ADVICE r5)
	if (ps_last_direct_) return launch_copy_two<T>(dst, pin_psN_dev_, err_count_, pin_psN_dev_ + ps_stride_, r_, stream_) == hipSuccess ? cnt : -1;
	// ONE small launch for both pieces (two runtime copies were two blit kernels)
	if (launch_copy_two<T>(dst, psN_, err_count_, psR_, r_, stream_) != hipSuccess) return -1;
	return cnt;
}

template <typename T>
void Engine<T>::finalize_error(bool resolve) {
	if (kl_pending_) {
		(void)hipEventSynchronize(err_event_);
		h_psN_.assign(pin_kl_, pin_kl_ + m_);                       // per-row terms of tr(H^T W^T V)
		h_klrow_.assign(pin_kl_ + m_, pin_kl_ + 2 * (size_t)m_);
		h_sW_.assign(pin_kl_ + 2 * (size_t)m_, pin_kl_ + 2 * (size_t)m_ + RP_);
		h_sH_.assign(pin_kl_ + 2 * (size_t)m_ + RP_, pin_kl_ + 2 * (size_t)m_ + 2 * RP_);
		h_psR_.assign(pin_kl_ + 2 * (size_t)m_ + 2 * RP_, pin_kl_ + 2 * (size_t)m_ + 2 * RP_ + r_);
		kl_pending_ = false;
		kl_unresolved_ = true;
	}
	if (resolve && kl_unresolved_) {
		resolve_error(h_vtv_, h_psN_, h_psR_, (long)((unsigned)m_ * (unsigned)(err_total_columns_ > 0 ? err_total_columns_ : n_)));
		double d = -sum_v_;
		for (int i = 0; i < m_; ++i) d += (double)h_klrow_[i];
		for (int c = 0; c < r_; ++c) d += (double)h_sW_[c] * (double)h_sH_[c];
		kl_ = d;
		kl_unresolved_ = false;
	}
	if (err_pending_) {
		(void)hipEventSynchronize(err_event_);
		h_psN_.assign(pin_psN_, pin_psN_ + err_count_);
		h_psR_.assign(pin_psR_, pin_psR_ + r_);
		err_pending_ = false;
		err_unresolved_ = true;
	}
	if (resolve && err_unresolved_) {
		resolve_error(h_vtv_, h_psN_, h_psR_, (long)((unsigned)m_ * (unsigned)(err_total_columns_ > 0 ? err_total_columns_ : n_)));
		err_unresolved_ = false;
	}
}

template <typename T>
void Engine<T>::resolve_error(const std::vector<T>& vtv_sorted, std::vector<T> htwtv, std::vector<T> hhtwtw, long total_elements) {
	frob2_ = resolve_frobenius_squared<T>(vtv_sorted, htwtv, hhtwtw);
	frob_ = std::sqrt(frob2_);
	rmsd_ = frob_ / std::sqrt((double)total_elements);
}

// ---- iteration --------------------------------------------------------------------------------

template <typename T>
Status Engine<T>::h_step(bool compute_error) {
	// first call of an iteration in the sharded form: decides whether this iteration's products are timed
	timing_now_ = timing_ && (timing_iter_++ % timing_stride_ == 0);
	if (prm_.divergence != 0) { kl_err_iter_ = compute_error; return kl_h_step(); }
	return h_step_impl(compute_error);
}

template <typename T>
Status Engine<T>::h_step_impl(bool compute_error) {
	const T eps = std::numeric_limits<T>::epsilon();
	h_product_ahead_ = false; f64_product_ahead_ = false;                           // (the three-phase API enqueues its own W^T V)
	if constexpr (std::is_same<T, float>::value) {
		if (fused_capable()) {
			// sharded form of the four-launch iteration (kernels_mu64.hip): K_H + U_H here
			if (!fused_ready_) {
				if (!gram_image_) HIPX(launch_mu64_gram_partials(Wt_, (int)mpad_, gramW_part_, stream_));
				normalize_next_ = 0;
				fused_ready_ = true;
			}
			GramReduceArgs rgW = gram_args(true, G_, scale_, normalize_next_);
			if (Status s = product_h(Wt_, &rgW, x3_ && wx3_valid_)) return s;
			if (Status s = mu64_update(false, slabs_, planH_.splits, slab_stride_, rgW.ksplit > 1 ? reinterpret_cast<const T*>(Gpart_) : G_, compute_error)) return s;
			hx3_valid_ = x3_;
			return ST_OK;
		}
	}
	hx3_valid_ = false;
	if constexpr (std::is_same<T, float>::value) {
		if (tri_) {
			// The fragments hold bf16(W) as the panel stores it -- unsmoothed, without the pending column scale D; the product's output gets both:
			// (W D S)^T V = S D (W^T V), applied by the H update to the summed slabs (PanelTriExtras).
			GramReduceArgs ride = {nullptr, 0, nullptr, nullptr, 0};
			if (Status s = tri_prepare_w(&ride)) return s;
			if (Status s = product_h(Wt_, ride.tri_frags != nullptr ? &ride : nullptr, true)) return s;
			T off, diag;
			tri_smoothing(&off, &diag);
			PanelTriExtras ex;
			ex.num_transform = true;
			if (tri_scale_pending_) { ex.num_colsq = colsq_; ex.num_colsq_parts = colsq_parts_; }
			ex.num_a = (float)(diag - off); ex.num_b = (float)off; ex.r = r_;
			// ... and the denominator S D G D S H from the Gram matrix as the reduction left it (its split image is in qx3_): D and S around the product
			ex.den_transform = true;
			ex.den_a = ex.num_a; ex.den_b = ex.num_b;
			ex.den_colsq = ex.num_colsq; ex.den_colsq_parts = ex.num_colsq_parts;
			// ... and leaves the operand of the next V (S H)^T behind: the bf16 fragments of the smoothed new columns (AlgorithmNonSmoothNMF.h:194)
			ex.frag_out = Hb_; ex.frag_KS = ksW_; ex.frag_a = (float)(diag - off); ex.frag_b = (float)off;
			// (old_as_bf16 stays off here: the H update is bound by what a CU can ingest, not by its MFMAs -- 28.3 -> 27.7 us -- and H's distance from the
			// fp64 oracle doubles, 5.7e-5 -> 1.5e-4 after 10 iterations; the W update gains 11 us at no measurable cost, tri_update_w)
			// (qx3_holds_g_: the Gram reduction left the split image of W^T W in qx3_ -- Q = nullptr tells the update kernel so; else it packs Gw_raw_)
			HIPX(launch_panel_update<T>(PANEL_MU, H_, slabs_, planH_.splits, slab_stride_, qx3_holds_g_ ? nullptr : reinterpret_cast<const T*>(Gw_raw_), RP_, (int)npad_, eps,
			                            compute_error ? psN_ : nullptr, n_, nullptr, nullptr, stream_, nullptr, nullptr, 0, qx3_, &ex));
			qx3_holds_g_ = false;
			hb_valid_ = true;
			return ST_OK;
		}
	}
	if (Status s = materialize_w()) return s;
	const T* F = Wt_;
	if (alg_ == ALG_NSNMF) {
		const T off = (T)prm_.theta / (T)(unsigned)r_;
		const T diag = (T)((1.0 - (T)prm_.theta) + off);
		HIPX(launch_smooth_panel<T>(Wt_, Ws_, RP_, r_, mpad_, off, diag, stream_));
		F = Ws_;
	}
	if (!(gram_w_ready_ && F == Wt_)) HIPX(launch_gram<T>(F, RP_, m_, gram_parts_, gram_part_, G_, stream_));
	gram_w_ready_ = false;      // consumed (the inverse passenger / update below read G_; W changes in the W step)
	const int S = planH_.splits;
	if (alg_ == ALG_MU || alg_ == ALG_NSNMF) {
		if (Status s = product_h(F, nullptr, x3_ && wx3_valid_ && F == Wt_)) return s;
		const bool emit = x3_ && alg_ == ALG_MU && panel_update_delivers_gram(RP_, sizeof(T));   // nsNMF's W step consumes the smoothed panel
		HIPX(launch_panel_update<T>(PANEL_MU, H_, slabs_, S, slab_stride_, G_, RP_, (int)npad_, eps,
		                            compute_error ? psN_ : nullptr, n_, nullptr, nullptr, stream_, nullptr, emit ? Hx3_ : nullptr, ksW_, qx3_));
		hx3_valid_ = emit;
	} else {
		T off = 0, diag = 0;
		if (alg_ == ALG_GDCLS) diag = (T)prm_.lambda;
		else if (alg_ == ALG_ACLS) diag = (T)prm_.lambdaH;
		else if (alg_ == ALG_AHCLS) {
			const T lam = (T)prm_.lambdaH, alpha = (T)prm_.alphaH;
			T beta = (T)((1 - alpha) * std::sqrt((double)(unsigned)r_) + alpha);
			beta *= beta;
			off = -lam; diag = lam * beta - lam;
		}
		if (compute_error) HIPX(hipMemcpyAsync(G2_, G_, sizeof(T) * (size_t)RP_ * RP_, hipMemcpyDeviceToDevice, stream_));
		// the inverse of the normal matrix (one workgroup) runs beside the product against V: as a passenger
		// workgroup of the product launch where that exists (fp32 MFMA product, r <= 64), else on the side stream
		if (inverse_rides(planH_)) {
			if constexpr (std::is_same<T, float>::value) {
				GramReduceArgs rg = {nullptr, 0, nullptr, nullptr, 0};
				rg.inv_a = G_; rg.inv_out = Qinv_; rg.inv_offdiag = off; rg.inv_diag = diag; rg.inv_r = r_;
				if (Status s = product_h(F, &rg, x3_ && wx3_valid_ && F == Wt_)) return s;
			}
		} else {
			if (Status s = normal_inverse_fork(G_, off, diag)) return s;
			if (Status s = product_h(F, nullptr, x3_ && wx3_valid_ && F == Wt_)) return s;
			if (Status s = normal_inverse_join()) return s;
		}
		const bool emit = x3_ && panel_update_delivers_gram(RP_, sizeof(T));
		T* hpart = nullptr;
		// (GDCLS on the split-operand path at padded rank 64: the fused iteration takes H H^T from the split image beside the product, Engine::iterate -- the
		//  sharded form, w_products(), makes its own Gram pass over H)
		if constexpr (std::is_same<T, float>::value) { if (gram_from_update() && !(alg_ == ALG_GDCLS && emit && RP_ == 64 && passengers_ride(planWx_) && h_partials_unneeded_)) hpart = gramH_part_; }
		HIPX(launch_panel_update<T>(PANEL_LS, H_, slabs_, S, slab_stride_, Qinv_, RP_, (int)npad_, eps,
		                            nullptr, n_, nullptr, nullptr, stream_, hpart, emit ? Hx3_ : nullptr, ksW_, qx3_));
		hx3_valid_ = emit;
		gram_h_partials_ = hpart != nullptr;
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::w_products(T* exchange) {
	if (prm_.divergence != 0) return kl_w_products(exchange, kl_err_iter_);      // (sparse Frobenius compute shards through the code below: its two products are SpMMs over the shard's images)
	T* ex_hht = exchange + (long)RP_ * mpad_;
	if constexpr (std::is_same<T, float>::value) {
		if (fused_capable()) {
			// K_W with the local H H^T reduced straight into the exchange buffer by the passenger
			// workgroups, then the local split-K slabs summed into the exchange panel
			GramReduceArgs rgH = gram_args(false, ex_hht, nullptr, 0);
			// a team of one: w_finish() sums the split-K slabs itself, as iterate_mu64() does -- no pass over the panel in between
			// (nsNMF: the H update of h_step left the split image of S H; without it -- no H step since the factors were set -- the smoothed panel is packed here)
			const T* Fh = H_;
			if (alg_ == ALG_NSNMF && !(x3_ && hx3_valid_)) {
				const T off = (T)prm_.theta / (T)(unsigned)r_;
				HIPX(launch_smooth_panel<T>(H_, Hs_, RP_, r_, npad_, off, (T)((1.0 - (T)prm_.theta) + off), stream_));
				Fh = Hs_;
			}
			if (sole_rank_ && gram_image_) return product_w(Fh, &rgH, nullptr, x3_ && hx3_valid_);
			// one K slice (short column shards) writes the exchange panel itself
			if (Status s = product_w(Fh, &rgH, exchange, x3_ && hx3_valid_)) return s;
			if (planW_.splits > 1) HIPX(launch_reduce_slabs<T>(slabs_, planW_.splits, slab_stride_, exchange, (long)RP_ * mpad_, stream_));
			return ST_OK;
		}
	}
	if constexpr (std::is_same<T, float>::value) {
		if (tri_) {
			GramReduceArgs ride = {nullptr, 0, nullptr, nullptr, 0};
			if (Status s = tri_prepare_h(ex_hht, sole_rank_, sole_rank_ ? &ride : nullptr)) return s;      // (a team of one: nothing is added to ex_hht before the W update reads it)
			if (Status s = product_w(H_, ride.tri_frags != nullptr ? &ride : nullptr, exchange, true)) return s;
			if (planW_.splits > 1) HIPX(launch_reduce_slabs<T>(slabs_, planW_.splits, slab_stride_, exchange, (long)RP_ * mpad_, stream_));
			return ST_OK;
		}
	}
	const T* Fh = H_;
	if (alg_ == ALG_NSNMF) {
		// the smoothed local columns S H_g enter both sums (AlgorithmNonSmoothNMF.h:194-197,213)
		const T off = (T)prm_.theta / (T)(unsigned)r_;
		const T diag = (T)((1.0 - (T)prm_.theta) + off);
		HIPX(launch_smooth_panel<T>(H_, Hs_, RP_, r_, npad_, off, diag, stream_));
		Fh = Hs_;
	}
	HIPX(launch_gram<T>(Fh, RP_, n_, gram_parts_, gram_part_, ex_hht, stream_));
	// a single K slice writes the exchange panel itself; several are summed into it
	if (Status s = product_w(Fh, nullptr, exchange)) return s;
	if (planW_.splits > 1) HIPX(launch_reduce_slabs<T>(slabs_, planW_.splits, slab_stride_, exchange, (long)RP_ * mpad_, stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::w_finish(const T* exchange, bool compute_error) {
	if (prm_.divergence != 0) return kl_w_finish(exchange, compute_error);
	const T eps = std::numeric_limits<T>::epsilon();
	const T* ex_hht = exchange + (long)RP_ * mpad_;
	if (alg_ != ALG_MU && alg_ != ALG_NSNMF) {
		// GDCLS / ALS / ACLS / AHCLS on the all-reduced sums (their extra products are r x r sized or row / column
		// separable): exactly the W step of iterate() with the reduced (V H^T)^T panel as the one "slab" and the reduced H H^T
		// (AlgorithmGradientDescentConstrainedLeastSquares.h:236-264, AlgorithmAlternatingHoyerConstrainedLeastSquares.h:226-284)
		const int norm_parts = panel_update_parts(RP_, sizeof(T), (int)mpad_);
		const bool ls_family = alg_ != ALG_GDCLS;
		if (compute_error) HIPX(launch_trace_small<T>(ex_hht, G2_, RP_, r_, psR_, stream_));      // G2_: W^T W saved before the regulariser (h_step)
		T* wpart = nullptr;
		if constexpr (std::is_same<T, float>::value) { if (gram_from_update()) wpart = gramW_part_; }
		wx3_valid_ = false;
		if (!ls_family) {
			HIPX(launch_panel_update<T>(PANEL_MU, Wt_, exchange, 1, 0, ex_hht, RP_, (int)mpad_, eps, nullptr, m_, sumsq_part_, compute_error ? numW_ : nullptr, stream_,
			                            wpart, nullptr, 0, qx3_));
			if (Status st = normalize_w(wpart != nullptr, norm_parts)) return st;
			// tr(H^T W^T V) as diag((V H^T)^T W) with the UPDATED W (GDCLS :259-264)
			if (compute_error) HIPX(launch_row_dot<T>(numW_, Wt_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
		} else {
			T offW = 0, diagW = 0;
			if (alg_ == ALG_ACLS) diagW = (T)prm_.lambdaW;
			else if (alg_ == ALG_AHCLS) {
				const T lam = (T)prm_.lambdaW, alpha = (T)prm_.alphaW;
				T beta = (T)((1 - alpha) * std::sqrt((double)(unsigned)r_) + alpha);
				beta *= beta;
				offW = -lam; diagW = lam * beta - lam;
			}
			// (the inverse destroys its input: a copy of the reduced H H^T, the exchange buffer stays the caller's)
			HIPX(hipMemcpyAsync(HHt_, ex_hht, sizeof(T) * (size_t)RP_ * RP_, hipMemcpyDeviceToDevice, stream_));
			if (Status st = normal_inverse(HHt_, offW, diagW)) return st;
			if (compute_error) HIPX(hipMemcpyAsync(Wold_, Wt_, sizeof(T) * (size_t)RP_ * mpad_, hipMemcpyDeviceToDevice, stream_));
			HIPX(launch_panel_update<T>(PANEL_LS, Wt_, exchange, 1, 0, Qinv_, RP_, (int)mpad_, eps, nullptr, m_, sumsq_part_, compute_error ? numW_ : nullptr, stream_,
			                            wpart, nullptr, 0, qx3_));
			// tr(W_old^T (V H^T)) over r diagonals (ALS :199-205)
			if (compute_error) HIPX(launch_row_dot<T>(Wold_, numW_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
			if (Status st = normalize_w(wpart != nullptr, norm_parts)) return st;
		}
		if (compute_error) { if (Status st = fetch_error_terms(r_)) return st; }
		return ST_OK;
	}
	if constexpr (std::is_same<T, float>::value) {
		if (fused_capable()) {
			// U_W on the all-reduced sums: one "slab" (the exchange panel), Q = the reduced H H^T  (a team of one: the split-K slabs as w_products left them)
			if (sole_rank_ && gram_image_) { if (Status s = mu64_update(true, slabs_, planW_.splits, slab_stride_, ex_hht, compute_error)) return s; }
			else if (Status s = mu64_update(true, exchange, 1, 0, ex_hht, compute_error)) return s;
			wx3_valid_ = x3_;
			normalize_next_ = 1;
			w_pending_ = true;
			if (compute_error) { if (Status s = fetch_error_terms(n_)) return s; }
			return ST_OK;
		}
	}
	if (compute_error) {
		const T* wtw = tri_ ? reinterpret_cast<const T*>(Gw_raw_) : G_;      // MU: W^T W of this iteration's H step (rank-256 bf16 path: the unscaled Gram matrix + tri_trace_scale())
		if (alg_ == ALG_NSNMF) {                            // unsmoothed W^T W (AlgorithmNonSmoothNMF.h:201-202)
			if (tri_) wtw = reinterpret_cast<const T*>(Gw_raw_);      // (by-product of this iteration's H step)
			else { HIPX(launch_gram<T>(Wt_, RP_, m_, gram_parts_, gram_part_, G2_, stream_)); wtw = G2_; }
		}
		HIPX(launch_trace_small<T>(ex_hht, wtw, RP_, r_, psR_, stream_, tri_trace_scale()));
		if (Status s = fetch_error_terms(n_)) return s;
	}
	wx3_valid_ = false;
	if (tri_) return tri_update_w(exchange, 1, 0, (sole_rank_ && qx3_holds_hht_) ? nullptr : ex_hht);
	HIPX(launch_panel_update<T>(PANEL_MU, Wt_, exchange, 1, 0, ex_hht, RP_, (int)mpad_, eps, nullptr, m_, sumsq_part_, nullptr, stream_, nullptr, nullptr, 0, qx3_));
	HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, panel_update_parts(RP_, sizeof(T), (int)mpad_), stream_));
	return ST_OK;
}

// W step of a column-sharded run whose ranks read each other's exchange buffers (comm.h, direct_exchange): the W update sums the ranks' (V_g H_g^T)^T panels in
// its prologue, in rank order -- every rank adds the same values in the same order, so the replicas of W stay bit-identical; the r x r parts are summed first
// (one small launch; nothing to do for a team of one).  Rank-64 multiplicative update on the split-operand path only (direct_w_finish()).
template <typename T>
Status Engine<T>::w_finish_peers(const T* const* exchanges, int count, bool compute_error) {
	if (!direct_w_finish() || count < 1 || count > PEER_SLABS_MAX || exchanges == nullptr) return ST_INVALID;
	if constexpr (std::is_same<T, float>::value) {
		PeerSlabs panels = {}, hhts = {};
		panels.count = hhts.count = count;
		for (int p = 0; p < count; ++p) { panels.p[p] = exchanges[p]; hhts.p[p] = exchanges[p] + (long)RP_ * mpad_; }
		const float* Q = hhts.p[0];
		if (count > 1) { HIPX(launch_sum_peers(hhts, HHt_, RP_ * RP_, stream_)); Q = HHt_; }
		if (Status s = mu64_update(true, nullptr, count, 0, Q, compute_error, &panels)) return s;
		wx3_valid_ = x3_;
		normalize_next_ = 1;
		w_pending_ = true;
		if (compute_error) { if (Status s = fetch_error_terms(n_)) return s; }
	}
	return ST_OK;
}

// ---- row-block form of the W step (one block per rank, see engine.h) -----------------------------------------------
template <typename T>
Status Engine<T>::w_update_rows(const T* num_rows, const T* hht, long row0, long rows, bool compute_error, T* colsq) {
	if (alg_ != ALG_MU && alg_ != ALG_NSNMF) return ST_INVALID;
	if (rows <= 0 || rows % 128 != 0 || row0 < 0 || row0 % 128 != 0 || row0 + rows > mpad_) return ST_INVALID;
	const T eps = std::numeric_limits<T>::epsilon();
	// Gw_raw_ is the Gram matrix of the panel WITHOUT its pending column scale: the trace below needs that scale even when materialize_w() is about to
	// fold it into the panel (a caller that used w_finish() before: ADVICE r3) -- the staged sums stay where they are
	const T* trace_scale = tri_trace_scale();
	if (Status s = materialize_w(false)) return s;      // (a pending column scale belongs to the old W; fold it in first.  This rank's rows are current: no gather)
	if (compute_error) {
		const T* wtw = tri_ ? reinterpret_cast<const T*>(Gw_raw_) : G_;      // MU: W^T W of this iteration's H step (rank-256 bf16 path: the unscaled Gram matrix + its scale)
		if (alg_ == ALG_NSNMF) {                            // unsmoothed W^T W (AlgorithmNonSmoothNMF.h:201-202)
			if (tri_) wtw = reinterpret_cast<const T*>(Gw_raw_);
			else { HIPX(launch_gram<T>(Wt_, RP_, m_, gram_parts_, gram_part_, G2_, stream_)); wtw = G2_; }
		}
		HIPX(launch_trace_small<T>(hht, wtw, RP_, r_, psR_, stream_, trace_scale));
		if (Status s = fetch_error_terms(n_)) return s;
	}
	const long valid = std::max<long>(0, std::min<long>(rows, (long)m_ - row0));
	HIPX(launch_panel_update<T>(PANEL_MU, Wt_ + row0 * RP_, num_rows, 1, 0, hht, RP_, (int)rows, eps, nullptr, (int)valid, sumsq_part_, nullptr, stream_,
	                            nullptr, nullptr, 0, qx3_));
	if constexpr (std::is_same<T, float>::value) {
		if (tri_) {
			HIPX(launch_colsq_stage(sumsq_part_, RP_, panel_update_parts(RP_, sizeof(T), (int)rows), colsq_, colsq, stream_));
			return ST_OK;
		}
	}
	HIPX(launch_reduce_partials<T>(sumsq_part_, panel_update_parts(RP_, sizeof(T), (int)rows), RP_, colsq, RP_, stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::w_normalize_rows(long row0, long rows, T* colsq) {
	if (rows <= 0 || rows % 128 != 0 || row0 < 0 || row0 + rows > mpad_) return ST_INVALID;
	// colsq: the r sums of squares over ALL rows (one "partial"): kernel::normalizeColumns' sum > 0 ? x / sqrt(sum) : x
	if constexpr (std::is_same<T, float>::value) {
		if (tri_) {
			// the same pass leaves the bf16 fragments of the normalised rows for the next W^T V (unsmoothed: S is applied to the product's output)
			HIPX(launch_finish_panel_bf16(Wt_, RP_, r_, row0, rows, colsq, colsq == colsq_ ? colsq_parts_ : 1, 0.0f, 1.0f, Wtb_, ksH_, stream_));
			tri_rows_cover_ = row0 == 0 && rows == mpad_;
			wtb_valid_ = tri_rows_cover_;
			tri_gw_ready_ = false;
			return ST_OK;
		}
	}
	HIPX(launch_normalize_panel_v2<T>(Wt_ + row0 * RP_, RP_, (int)rows, colsq, 1, stream_));
	return ST_OK;
}

// ---- padded rank 256, bf16 product operands (kernels_tri.hip) ---------------------------------------------------------------------
template <typename T>
void Engine<T>::tri_smoothing(T* offdiag, T* diag) const {
	if (alg_ == ALG_NSNMF) {
		*offdiag = (T)prm_.theta / (T)(unsigned)r_;
		*diag = (T)((1.0 - (T)prm_.theta) + *offdiag);
	} else { *offdiag = 0; *diag = 1; }
}

template <typename T>
Status Engine<T>::tri_prepare_w(GramReduceArgs* ride) {
	if constexpr (std::is_same<T, float>::value) {
		const bool wide = qx3_ != nullptr && panel_update_wide_available(RP_);
		if (!wtb_valid_) {
			if (Status s = ensure_w_rows()) return s;      // (the fp32 rows of the other ranks first: engine.h, w_fragment_exchange)
			// plain re-rounding of the panel as it lies (no smoothing, no normalisation: W is as the caller / the gather left it)
			HIPX(launch_finish_panel_bf16(Wt_, RP_, r_, 0, mpad_, nullptr, 0, 0.0f, 1.0f, Wtb_, ksH_, stream_));
			wtb_valid_ = true;
		}
		if (!tri_gw_ready_) {
			// W^T W of the ROUNDED W (the matrix the product multiplies V with), as the reduction leaves it: Gw_raw_ and its split image are NOT normalised
			// and not smoothed -- the H update applies D and S around its r x r product (PanelTriExtras::den_transform), the error term's trace applies D
			// (AlgorithmNonSmoothNMF.h:201-202 wants the unsmoothed W^T W).  Round 3 first ran k_smooth_gram here (6.5 us) and a staging launch for D (4.7 us).
			// The diagonal of the matrix IS the new pending scale: sums of squares of the rounded columns (one "staged" vector).
			if (ride != nullptr && tri_ride_w_) {
				// as passengers of the W^T V launch the caller is about to make: every consumer (H update, error trace, the next W update's pending scale) runs behind it
				ride->tri_frags = Wtb_; ride->tri_ks = ksH_; ride->tri_partial = gram_tri_part_; ride->tri_counters = tri_ride_counters_; ride->G = Gw_raw_; ride->tri_x3 = wide ? qx3_ : nullptr; ride->tri_diag = tri_scale_from_gram_ ? colsq_ : nullptr;
			} else HIPX(launch_gram_tri_bf16_image(Wtb_, RP_, ksH_, num_cus_, gram_tri_part_, Gw_raw_, wide ? qx3_ : nullptr, tri_scale_from_gram_ ? colsq_ : nullptr, num_cus_, stream_));
			if (tri_scale_from_gram_) { colsq_parts_ = 1; tri_scale_from_gram_ = false; }
			tri_gw_ready_ = true;
			qx3_holds_g_ = wide;
		}
	}
	return ST_OK;
}

// hht = (S H)(S H)^T for the W step, from the bf16 fragments of the smoothed columns the H update left in Hb_ -- the Gram matrix of exactly the operand
// V is multiplied with.  local_q: hht stays what the W update multiplies with (no reduction over ranks in between), so the reduction also leaves its split
// image in qx3_ and the update's launcher need not pack it (k_pack_panel_x3, 4.9 us).
template <typename T>
Status Engine<T>::tri_prepare_h(T* hht, bool local_q, GramReduceArgs* ride) {
	if constexpr (std::is_same<T, float>::value) {
		if (!hb_valid_) {
			// H did not come from this engine's H update (set_factors / randomize between the two half-steps): one finishing pass makes the fragments
			T off, diag;
			tri_smoothing(&off, &diag);
			HIPX(launch_finish_panel_bf16(H_, RP_, r_, 0, npad_, nullptr, 0, off, diag, Hb_, ksW_, stream_));
			hb_valid_ = true;
		}
		const bool image = local_q && qx3_ != nullptr && panel_update_wide_available(RP_);
		if (image && ride != nullptr && tri_ride_h_) { ride->tri_frags = Hb_; ride->tri_ks = ksW_; ride->tri_partial = gram_tri_part_; ride->tri_counters = tri_ride_counters_; ride->G = hht; ride->tri_x3 = qx3_; ride->tri_diag = nullptr; }      // (rides in the V (S H)^T launch)
		else if (image) HIPX(launch_gram_tri_bf16_image(Hb_, RP_, ksW_, num_cus_, gram_tri_part_, hht, qx3_, nullptr, num_cus_, stream_));
		else HIPX(launch_gram_tri_bf16(Hb_, RP_, ksW_, num_cus_, gram_tri_part_, hht, nullptr, 0, num_cus_, stream_));
		qx3_holds_g_ = false;
		qx3_holds_hht_ = image;
	}
	return ST_OK;
}

// The W update of the rank-256 bf16 path on the whole panel: num = the reduced (V (S H)^T)^T panel (S slabs), hht its r x r operand
// (nullptr: its split image is in qx3_).  ONE pass over the panel: the update kernel reads the old rows with their pending column scale, writes the new
// rows unnormalised, their bf16 fragments (what the next W^T V multiplies with) and the partial sums of squares; the column normalisation
// (kernel::normalizeColumns, KernelNormalizeColumns.cu:37-58) becomes the new pending scale.  Round 2 followed the update with k_finish_panel_bf16
// (read + write of the panel, 30 us at config 4) and took the Gram matrix from the fp32 panel by split operands (k_gram_tri_x3, 29 us).
template <typename T>
Status Engine<T>::tri_update_w(const T* num, int S, long stride, const T* hht) {
	if constexpr (std::is_same<T, float>::value) {
		const T eps = std::numeric_limits<T>::epsilon();
		const int parts = panel_update_parts(RP_, sizeof(T), (int)mpad_);
		wx3_valid_ = false;
		PanelTriExtras ex;
		if (tri_scale_pending_) { ex.old_colsq = colsq_; ex.old_colsq_parts = colsq_parts_; }
		ex.frag_out = Wtb_; ex.frag_KS = ksH_;
		// the old rows enter W (SH)(SH)^T rounded to bf16: 63.7 -> 52.3 us, W's distance from the fp64 oracle unchanged (3.4e-4 vs 3.2e-4 after 10 iterations,
		// 1.17e-3 vs 1.16e-3 after 40, tools/tri_accuracy.py); NMFAMD_TRI_FP32_DEN=1 keeps the six-term product
		ex.old_as_bf16 = tri_w_den_bf16_;
		HIPX(launch_panel_update<T>(PANEL_MU, Wt_, num, S, stride, hht, RP_, (int)mpad_, eps, nullptr, m_, nullptr, nullptr, stream_, nullptr, nullptr, 0, qx3_, &ex));
		qx3_holds_g_ = qx3_holds_hht_ = false;
		// The new pending scale comes out of the next Gram reduction (tri_prepare_w): d(c) = 1 / sqrt(sum of squares of the ROUNDED column c) -- the column
		// norms of the matrix the product multiplies with; they differ from the fp32 norms by ~1e-5 relative (zero-mean rounding over m rows).  Every consumer
		// (H update, next W update, error trace, materialize_w) runs after that reduction.
		tri_scale_pending_ = true;
		tri_scale_from_gram_ = true;
		(void)parts;
		wtb_valid_ = true;
		tri_rows_cover_ = true;
		tri_gw_ready_ = false;
		return ST_OK;
	}
	return ST_INVALID;
}

template <typename T>
bool Engine<T>::fused_capable() const {
	// (evaluated before allocate(): no tiled_ here.  nsNMF (round 6): only on the split-operand products with the Gram matrices taken from the images -- x3_ is false
	//  until the plan has chosen that path, so the one-pass request and the native-fp32 plan never see it)
	return std::is_same<T, float>::value && RP_ == 64 &&
	       (alg_ == ALG_MU || (alg_ == ALG_NSNMF && x3_ && !bf16_ && !sparse_ && std::getenv("NMFAMD_GRAM_PARTIALS") == nullptr)) &&
	       std::getenv("NMFAMD_FORCE_VALU") == nullptr && std::getenv("NMFAMD_NO_FUSED_MU") == nullptr;
}

// fp32 at padded ranks 128 ... 512 on the split-operand products (ranks 65 ... 512 in float: what rounds 1 - 5 ran as the generic sequence of 14 launches):
// multiplicative update and nsNMF.  The bf16 mode keeps its own path.
template <typename T>
bool Engine<T>::fused32w_capable() const {
	return std::is_same<T, float>::value && x3_ && tiled_ && !bf16_ && !sparse_ && (alg_ == ALG_MU || alg_ == ALG_NSNMF) && RP_ >= 128 && panel_update_wide_available(RP_) &&
	       gram_wide_available(RP_) && qx3_ != nullptr &&
	       std::getenv("NMFAMD_FORCE_VALU") == nullptr && std::getenv("NMFAMD_NO_FUSED_MU") == nullptr;
}

// One iteration in EIGHT launches, none of them a pack, a smoothing pass or a normalisation (the generic sequence: 14.2 at r = 128):
//   Gram slices of Wt (k_gram_wide_x3) -> k_gram_reduce_x3: Wt^T Wt, ITS split image (the update's operand) and the pending column scale d
//   Wt^T V from the split image the W update left
//   H update (FX): num <- S D (sum of the slabs), den = S D (Wt^T Wt) D S h around the MFMA product; writes H, S H and the split image of the next product's operand
//   the same for S H: Gram slices -> reduction + split image; V (S H)^T; W update (old rows read as Wt d, result unnormalised, its split image, sums of squares)
// The float counterpart of iterate_fused64 without passengers: the split-operand product kernel is the headline's and stays as it is.
template <typename T>
Status Engine<T>::iterate_fused32w(bool compute_error) {
	if constexpr (std::is_same<T, float>::value) {
		const float eps = std::numeric_limits<float>::epsilon();
		const bool ns = alg_ == ALG_NSNMF;
		const float off = ns ? (float)prm_.theta / (float)(unsigned)r_ : 0.f;
		const float diag = ns ? (float)((1.0 - (float)prm_.theta) + off) : 1.f;
		const int norm_parts = panel_update_parts(RP_, sizeof(T), (int)mpad_);
		const float* dscale = f32w_pending_ ? f32w_scale_ : nullptr;
		// H step
		if (f32w_ride_h_ > 0) {
			GramReduceArgs rg = {nullptr, 0, nullptr, nullptr, 0};
			rg.wide_P = Wt_; rg.wide_len = m_; rg.wide_parts = f32w_ride_h_; rg.wide_partial = gram_part_;
			if (Status s = product_h(Wt_, &rg, wx3_valid_)) return s;
			HIPX(launch_gram_reduce_x3(gram_part_, f32w_ride_h_, RP_, G_, qx3_, f32w_pending_ ? sumsq_part_ : nullptr, norm_parts, f32w_scale_, stream_));
		} else {
			HIPX(launch_gram_wide_fused_f32(Wt_, RP_, m_, gram_parts_, gram_part_, G_, qx3_, f32w_pending_ ? sumsq_part_ : nullptr, norm_parts, f32w_scale_, stream_));
			if (Status s = product_h(Wt_, nullptr, wx3_valid_)) return s;
		}
		wx3_valid_ = true;
		PanelFusedF32 fh;
		fh.h_side = 1; fh.scale = dscale; fh.r = r_;
		if (ns) { fh.smooth = 1; fh.off = off; fh.diag = diag; fh.smooth_out = Hs_; }
		fh.x3_out = Hx3_; fh.x3_ks = ksW_;
		HIPX(launch_panel_update_wide_f32(PANEL_MU, H_, slabs_, planH_.splits, slab_stride_, nullptr, RP_, (int)npad_, eps, compute_error ? psN_ : nullptr, n_, nullptr, nullptr,
		                                  stream_, qx3_, nullptr, &fh));
		// W step
		const float* Fh = ns ? Hs_ : H_;
		if (f32w_ride_w_ > 0) {
			GramReduceArgs rg = {nullptr, 0, nullptr, nullptr, 0};
			rg.wide_P = Fh; rg.wide_len = n_; rg.wide_parts = f32w_ride_w_; rg.wide_partial = gram_part_;
			if (Status s = product_w(Fh, &rg, nullptr, true)) return s;
			HIPX(launch_gram_reduce_x3(gram_part_, f32w_ride_w_, RP_, HHt_, qx3_, nullptr, 0, nullptr, stream_));
		} else {
			HIPX(launch_gram_wide_fused_f32(Fh, RP_, n_, gram_parts_, gram_part_, HHt_, qx3_, nullptr, 0, nullptr, stream_));
			if (Status s = product_w(Fh, nullptr, nullptr, true)) return s;
		}
		hx3_valid_ = !ns;
		// tr((S H)(S H)^T W^T W) with the W^T W of this iteration's H step (the pending scale applied on the way: D G D)
		if (compute_error) HIPX(launch_trace_small<T>(HHt_, G_, RP_, r_, psR_, stream_, nullptr, dscale));
		PanelFusedF32 fw;
		fw.old_scale = dscale;
		fw.x3_out = Wx3_; fw.x3_ks = ksH_;
		HIPX(launch_panel_update_wide_f32(PANEL_MU, Wt_, slabs_, planW_.splits, slab_stride_, nullptr, RP_, (int)mpad_, eps, nullptr, m_, sumsq_part_, nullptr,
		                                  stream_, qx3_, nullptr, &fw));
		f32w_pending_ = true;
		wx3_valid_ = true;          // (of the panel as it lies)
		gram_w_ready_ = false;
		if (compute_error) { if (Status s = fetch_error_terms(n_)) return s; }
	}
	return ST_OK;
}

// Double precision, multiplicative update and nsNMF on the MFMA kernels of kernels_f64.hip at any padded rank they cover: what the reference's own callers run
// (example/main.cpp: NmfDescription<double>, nsNMF, r = 158; the R binding).  NMFAMD_NO_FUSED_MU=1 keeps the generic launch sequence (the cross-check path).
template <typename T>
bool Engine<T>::fused64_capable() const {
	return std::is_same<T, double>::value && tiled_ && !sparse_ && !bf16_ && (alg_ == ALG_MU || alg_ == ALG_NSNMF) && RP_ % 64 == 0 && RP_ <= 512 &&
	       (RP_ == 64 || panel_update_wide_f64_available(RP_)) && std::getenv("NMFAMD_FORCE_VALU") == nullptr && std::getenv("NMFAMD_NO_FUSED_MU") == nullptr;
}

// One iteration in FOUR launches (the generic sequence: 10 at padded rank 64, 12 for nsNMF -- smoothing x 2, Gram + reduction x 2, product x 2, update x 2,
// normalise + compaction of its partial sums):
//   1. W^T V from the panel as it lies (Wt: unnormalised, unsmoothed) + passengers: Wt^T Wt, and the column scale d from the last W update's sums of squares
//   2. H update: numerator S D (sum of the slabs) -- (Wt D S)^T V = S D (Wt^T V) --, denominator S D (Wt^T Wt) D S h with D and S applied around the MFMA
//      product; writes H and, nsNMF, the smoothed panel S H
//   3. V (S H)^T + passengers: (S H)(S H)^T
//   4. W update: old rows read as W d, result left unnormalised + per-workgroup sums of squares (the first half of kernel::normalizeColumns)
// References: AlgorithmMultiplicativeFrobenius.h:165-248, AlgorithmNonSmoothNMF.h:174-218 (same operation order per element; the column scale is a factor
// 1 / sqrt(sum) where the reference divides by sqrt(sum): 1 ulp, inside the 1e-9 the fp64 tests ask).
// launch 1 of the fused double-precision iteration: W^T V from the panel as it lies + the Gram passengers (W^T W, the pending column scale)
template <typename T>
Status Engine<T>::fused64_product_h() {
	if constexpr (std::is_same<T, double>::value) {
		struct RideGuard { const GramRideF64*& p; ~RideGuard() { p = nullptr; } } guard{ride64_};
		GramRideF64 gw = {};
		gw.P = Wt_; gw.len = m_; gw.slices = f64_slices_h_; gw.partial = f64_partial_; gw.counters = f64_counters_; gw.G = G_;
		gw.sumsq_part = f64_pending_ ? sumsq_part_ : nullptr; gw.sumsq_parts = panel_update_parts(RP_, sizeof(T), (int)mpad_); gw.scale_out = f64_scale_;
		if (const char* e = tuning_env("NMFAMD_RIDE64_STOP")) gw.stop = std::atoi(e);
		gw.stamps = f64_stamps_;
		gw.items = f64_items_h_;
		ride64_ = &gw;
		return product_h(Wt_);
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::iterate_fused64(bool compute_error) {
	if constexpr (std::is_same<T, double>::value) {
		const double eps = std::numeric_limits<double>::epsilon();
		const bool ns = alg_ == ALG_NSNMF;
		const double off = ns ? prm_.theta / (double)(unsigned)r_ : 0.0;
		const double diag = ns ? (1.0 - prm_.theta) + off : 1.0;
		struct RideGuard { const GramRideF64*& p; ~RideGuard() { p = nullptr; } } guard{ride64_};
		// Error iterations of a single engine: the H update writes its n terms and the W update's trace workgroups their r terms straight into the pinned host
		// buffer (its device address): no k_trace_small launch, no copy launch (as iterate_mu64 does since round 5) -- 1.2 us per iteration on average at the
		// reference example's shape.  The previous error iteration's values leave the buffer first.
		ps_direct_ = compute_error && !error_terms_stay_ && pin_psN_dev_ != nullptr && tuning_env("NMFAMD_ERROR_COPY_KERNEL") == nullptr;
		if (ps_direct_) finalize_error(false);
		struct DirectGuard { bool& f; ~DirectGuard() { f = false; } } direct_guard{ps_direct_};
		T* const ps_n = ps_direct_ ? pin_psN_dev_ : psN_;
		T* const ps_r = ps_direct_ ? pin_psN_dev_ + ps_stride_ : psR_;
		// 1 (already enqueued by begin_next_iteration() behind the previous error iteration: nothing has touched W, V or the scratch since)
		if (f64_product_ahead_) f64_product_ahead_ = false;
		else if (Status s = fused64_product_h()) return s;
		// 2
		PanelFusedF64 fh = {};
		const double* dscale = f64_pending_ ? f64_scale_ : nullptr;
		fh.h_side = (dscale != nullptr || ns) ? 1 : 0;      // (a normalised W and no smoothing: G_ is W^T W as the update needs it)
		fh.scale = dscale; fh.r = r_;
		if (ns) { fh.smooth = 1; fh.off = off; fh.diag = diag; fh.smooth_out = Hs_; }
		fh.stamps = f64_stamps_ != nullptr ? f64_stamps_ + 1l * 4096 * 8 : nullptr;
		if (RP_ == 64) HIPX(launch_panel_update64_f64(PANEL_MU, H_, slabs_, planH_.splits, slab_stride_, G_, (int)npad_, eps, compute_error ? ps_n : nullptr, n_, nullptr, nullptr, stream_, &fh));
		else HIPX(launch_panel_update_wide_f64(PANEL_MU, H_, slabs_, planH_.splits, slab_stride_, G_, RP_, (int)npad_, eps, compute_error ? ps_n : nullptr, n_, nullptr, nullptr, stream_, &fh));
		// 3
		const double* Fh = ns ? Hs_ : H_;
		GramRideF64 gh = {};
		gh.P = Fh; gh.len = n_; gh.slices = f64_slices_w_; gh.partial = f64_partial_; gh.counters = f64_counters_; gh.G = HHt_;
		if (const char* e = tuning_env("NMFAMD_RIDE64_STOP")) gh.stop = std::atoi(e);
		gh.stamps = f64_stamps_ != nullptr ? f64_stamps_ + 2l * 4096 * 8 : nullptr;
		gh.items = f64_items_w_;
		ride64_ = &gh;
		if (Status s = product_w(Fh)) return s;
		ride64_ = nullptr;
		// 4 -- and, on error iterations, tr((S H)(S H)^T W^T W) with the W^T W of this iteration's H step, unsmoothed (AlgorithmNonSmoothNMF.h:201-202;
		// AlgorithmMultiplicativeFrobenius.h:212) by extra workgroups of the same launch (G_ is the Gram matrix of the panel as it lies: the pending scale is applied
		// on the way, D G D = W^T W of the normalised W)
		PanelFusedF64 fw = {};
		fw.old_scale = dscale;
		if (compute_error) { fw.trace_a = HHt_; fw.trace_b = G_; fw.trace_scale = dscale; fw.trace_out = ps_r; fw.trace_r = r_; }
		fw.stamps = f64_stamps_ != nullptr ? f64_stamps_ + 3l * 4096 * 8 : nullptr;
		if (RP_ == 64) HIPX(launch_panel_update64_f64(PANEL_MU, Wt_, slabs_, planW_.splits, slab_stride_, HHt_, (int)mpad_, eps, nullptr, m_, sumsq_part_, nullptr, stream_, &fw));
		else HIPX(launch_panel_update_wide_f64(PANEL_MU, Wt_, slabs_, planW_.splits, slab_stride_, HHt_, RP_, (int)mpad_, eps, nullptr, m_, sumsq_part_, nullptr, stream_, &fw));
		f64_pending_ = true;
		gram_w_ready_ = false;
		hx3_valid_ = false; wx3_valid_ = false;
		if (compute_error) { if (Status s = fetch_error_terms(n_)) return s; }
	}
	return ST_OK;
}

// GDCLS and the ALS family never smooth their factors, so the Gram matrix the next half-step needs is the Gram matrix of
// exactly what the update kernel has just written: at fp32 / padded rank 64 that kernel emits it as partial matrices
// (64 panel rows each) and the 16-block reduction of the fused MU path replaces a pass over the panel.
template <typename T>
bool Engine<T>::gram_from_update() const {
	return std::is_same<T, float>::value && alg_ >= ALG_GDCLS && alg_ <= ALG_AHCLS && panel_update_delivers_gram(RP_, sizeof(T)) &&
	       tuning_env("NMFAMD_NO_GRAM_FROM_UPDATE") == nullptr;
}

// Column normalisation of W after its update (kernel::normalizeColumns).  With partial Gram matrices of the unnormalised W
// at hand: their reduction yields the column scales 1 / ||W(:, c)|| and W^T W of the NORMALISED W in one 16-block launch,
// W is scaled in place, and the next H step finds its Gram matrix ready.  Otherwise: norms from the sum-of-squares partials.
template <typename T>
Status Engine<T>::normalize_w(bool from_gram_partials, int norm_parts) {
	if constexpr (std::is_same<T, float>::value) {
		if (from_gram_partials) {
			if (x3_ && Graw64_ != nullptr && tuning_env("NMFAMD_NORMALIZE_TWO_LAUNCHES") == nullptr) {
				// ONE launch (round 4): column scales from the update's sums of squares, G = D (sum of the partial Gram matrices) D, W <- W D and its split image
				HIPX(launch_gram64_reduce_scale_all(gramW_part_, (int)(mpad_ / 64), sumsq_part_, norm_parts, G_, scale_, Wt_, (int)mpad_, Wx3_, ksH_, stream_));
			} else if (x3_ && Graw64_ != nullptr) {
				// reduction, then ONE launch: scales from the raw diagonal, G = D Graw D, W <- W D and its split image for the next W^T V
				HIPX(launch_gram64_normalize_all(gramW_part_, (int)(mpad_ / 64), Graw64_, G_, scale_, Wt_, (int)mpad_, Wx3_, ksH_, stream_));
			} else {
			HIPX(launch_gram64_from_partials(gramW_part_, (int)(mpad_ / 64), G_, scale_, stream_));
			// the scaling pass also leaves the split image of the normalised W for the next W^T V
			HIPX(launch_mu64_apply_scale(Wt_, (int)mpad_, scale_, stream_, x3_ ? Wx3_ : nullptr, ksH_));
			}
			wx3_valid_ = x3_;
			gram_w_ready_ = true;
			return ST_OK;
		}
	}
	gram_w_ready_ = false;
	HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, norm_parts, stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::ensure_w_rows() {
	// (row-block sharded run with the fragment exchange: whoever wants the WHOLE fp32 panel gets the other ranks' rows first -- a collective, engine.h)
	if (w_rows_stale_) {
		if (!w_gather_hook_) { last_error_ = "the fp32 rows of the other ranks' blocks were not gathered (the sharded run that owns the exchange is closed)"; return ST_INVALID; }
		if (Status s = w_gather_hook_()) return s;
		w_rows_stale_ = false;
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::materialize_w(bool whole_panel) {
	if (whole_panel) { if (Status s = ensure_w_rows()) return s; }
	if (f32w_pending_) {
		// the fused fp32 iteration at padded ranks >= 128 left W unnormalised (and the split image of THAT panel in Wx3_)
		HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, panel_update_parts(RP_, sizeof(T), (int)mpad_), stream_));
		f32w_pending_ = false;
		wx3_valid_ = false;
		gram_w_ready_ = false;
	}
	if (kl_scale_pending_) {
		// the KL iteration left W unnormalised (iterate_kl): the pass it skipped
		HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, (int)(mpad_ / 128), stream_));
		kl_scale_pending_ = false;
	}
	if (f64_pending_) {
		// the fused double-precision iteration left W unnormalised: the second half of kernel::normalizeColumns, as the generic iteration runs it after every W update
		HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, panel_update_parts(RP_, sizeof(T), (int)mpad_), stream_));
		f64_pending_ = false;
		f64_product_ahead_ = false;      // (an ahead launch multiplied with the unnormalised panel and its pending scale)
		gram_w_ready_ = false;
	}
	if constexpr (std::is_same<T, float>::value) {
		if (w_pending_) {
			// column norms from the partial Grams of the unnormalised W, then W <- W diag(scale)
			GramReduceArgs rg = gram_args(true, G2_, scale_, 1);
			if (Status s = standalone_gram(rg)) return s;
			HIPX(launch_mu64_apply_scale(Wt_, (int)mpad_, scale_, stream_));
			w_pending_ = false;
			fused_ready_ = false;
			h_product_ahead_ = false; f64_product_ahead_ = false;
			wx3_valid_ = false;
		}
		if (tri_scale_pending_) {
			if (tri_scale_from_gram_) { if (Status s = tri_prepare_w()) return s; }      // (the scale of the last W update comes out of the Gram reduction)
			// rank-256 bf16 path: W <- W diag(d), d from the staged sums of squares; the fragments (of the unscaled panel) are re-made on demand, the Gram matrices already
			// describe the normalised W
			HIPX(launch_scale_panel_tri(Wt_, RP_, mpad_, colsq_, colsq_parts_, nullptr, stream_));
			tri_scale_pending_ = false;
			wtb_valid_ = false;
			tri_gw_ready_ = false;        // (Gw_raw_ and its image describe the unscaled panel)
			qx3_holds_g_ = false;
		}
	}
	return ST_OK;
}

// One pass over V: G = W^T W (split image of W), then ONE persistent launch for W^T V, the H update and V H^T, then U_W.
template <typename T>
Status Engine<T>::iterate_onepass(bool compute_error) {
#ifdef NMFAMD_DIAG_BUILD
	if constexpr (std::is_same<T, float>::value) {
		if (!fused_ready_) { normalize_next_ = 0; fused_ready_ = true; }
		if (!wx3_valid_) { HIPX(launch_pack_panel_x3(Wt_, RP_, m_, Wx3_, ksH_, stream_)); wx3_valid_ = true; }
		GramReduceArgs rgW = gram_args(true, G_, scale_, normalize_next_);
		rgW.ksplit = 0; rgW.spread = 0; rgW.G = G_;       // (the persistent launch reads the finished matrix)
		if (Status s = standalone_gram(rgW)) return s;
		OnePassArgs a;
		a.V = V_; a.tile_stride = strideV_;
		a.Wx3 = Wx3_; a.G = G_; a.scale = scale_; a.ps = op_ps4_; a.ps_stride = npad_;
		if constexpr (std::is_same<T, float>::value) { a.H = H_; a.H_out = op_H2_; }
		a.slabs = op_slabs_; a.slab_stride = (long)RP_ * mpad_;
		a.hh_part = op_hh_part_; a.part_scratch = op_part_; a.hfrag_scratch = op_hfrag_;
		a.ticket = op_ctl_; a.abort_flag = op_ctl_ + 8;
		a.tile_rows = (int)(mpad_ / 16); a.w_ks = ksH_; a.n = n_; a.panels = (n_ + 31) / 32;
		a.seq = op_seq_++;
		a.compute_error = compute_error ? 1 : 0;
		a.eps = std::numeric_limits<float>::epsilon();
		a.stamps = op_stamps_;
		record_begin();
#ifdef NMFAMD_DIAG_BUILD
		HIPX(launch_mu64_onepass(a, stream_));
#else
		return ST_INVALID;      // (one_pass_ is never set in the shipped library)
#endif
		record_end();
		std::swap(H_, op_H2_);                             // the new H
		HIPX(launch_reduce_partials<float>(op_hh_part_, ONEPASS_XCDS * ONEPASS_GROUP, 4096, HHt_, 4096, stream_));
		if (compute_error) HIPX(launch_reduce_partials<float>(op_ps4_, 4, npad_, psN_, n_, stream_));   // the four owner waves' parts of the per-column terms
		if (Status s = mu64_update(true, op_slabs_, ONEPASS_XCDS, (long)RP_ * mpad_, HHt_, compute_error)) return s;
		wx3_valid_ = x3_;
		hx3_valid_ = false;
		normalize_next_ = 1;
		w_pending_ = true;
		if (compute_error) {
			if (Status s = fetch_error_terms(n_)) return s;
			if (Status s = onepass_check()) return s;
		}
	}
#else
	(void)compute_error;
#endif
	return ST_OK;
}

// A launch that could not form its groups (an XCD with other than 32 of its workgroups: another kernel held CUs) gives
// up after bounded waits and leaves the abort word set; its outputs are void.
template <typename T>
Status Engine<T>::onepass_check() {
#ifdef NMFAMD_DIAG_BUILD

	HIPX(hipMemcpyAsync(pin_abort_, op_ctl_ + 8, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
	HIPX(hipStreamSynchronize(stream_));
	if (op_stamps_ != nullptr) {
		// diagnostic builds: the stamps of the last launch go to the file NMFAMD_ONEPASS_STAMPS names
		std::vector<unsigned long long> h(16 * 8 * ONEPASS_XCDS * ONEPASS_GROUP);
		if (hipMemcpy(h.data(), op_stamps_, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
			if (FILE* f = std::fopen(tuning_env("NMFAMD_ONEPASS_STAMPS"), "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); }
		}
	}
	if (*pin_abort_ != 0) {
		one_pass_ = false; one_pass_gave_up_ = true;
		last_error_ = "the one-pass iteration could not keep its workgroups resident (another kernel on the device?): factors are void; unset NMFAMD_ONE_PASS";
		return ST_HIP_ERROR;
	}
#endif
	return ST_OK;
}

template <typename T>
Status Engine<T>::iterate_mu64(bool compute_error) {
	if constexpr (std::is_same<T, float>::value) {
		if (one_pass_) return iterate_onepass(compute_error);
		const float eps = std::numeric_limits<float>::epsilon();
		// Error iterations of a single engine: the two update kernels write their n + r terms into the pinned host buffer themselves (its device address): the
		// copy launch behind them (k_copy_small: 4.1 us, every tenth iteration) goes.  The previous error iteration's values leave the buffer first.
		static const bool no_direct = tuning_env("NMFAMD_ERROR_MEMCPY") != nullptr || tuning_env("NMFAMD_ERROR_COPY_KERNEL") != nullptr;
		ps_direct_ = compute_error && gram_image_ && !error_terms_stay_ && pin_psN_dev_ != nullptr && !no_direct;
		if (ps_direct_) finalize_error(false);
		struct DirectGuard { bool& f; ~DirectGuard() { f = false; } } direct_guard{ps_direct_};
		if (!fused_ready_) {
			if (!gram_image_) HIPX(launch_mu64_gram_partials(Wt_, (int)mpad_, gramW_part_, stream_));
			normalize_next_ = 0;
			fused_ready_ = true;
		}
		GramReduceArgs rgW = gram_args(true, G_, scale_, normalize_next_);
		if (h_product_ahead_) h_product_ahead_ = false;      // (begin_next_iteration() enqueued exactly this launch)
		else if (Status s = product_h(Wt_, &rgW, x3_ && wx3_valid_)) return s;
		if (Status s = mu64_update(false, slabs_, planH_.splits, slab_stride_, rgW.ksplit > 1 ? reinterpret_cast<const T*>(Gpart_) : G_, compute_error)) return s;
		GramReduceArgs rgH = gram_args(false, HHt_, nullptr, 0);
		if (Status s = product_w(H_, &rgH, nullptr, x3_)) return s;
		if (Status s = mu64_update(true, slabs_, planW_.splits, slab_stride_, HHt_, compute_error)) return s;
		wx3_valid_ = x3_;
		hx3_valid_ = false;
		normalize_next_ = 1;
		w_pending_ = true;
		if (compute_error) {
			if (Status s = fetch_error_terms(n_)) return s;
		}
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::begin_next_iteration() {
	if constexpr (std::is_same<T, double>::value) {
		// the fused double-precision iteration: its first launch writes the split-K slabs, G_ and the scale vector -- scratch the next iterate() overwrites anyway
		if (!fused64_capable() || f64_partial_ == nullptr || !f64_pending_ || f64_product_ahead_ || prm_.divergence != 0) return ST_OK;
		const bool sampled = timing_now_;
		timing_now_ = false;
		const Status s = fused64_product_h();
		timing_now_ = sampled;
		if (s != ST_OK) return s;
		f64_product_ahead_ = true;
		return ST_OK;
	}
	if constexpr (std::is_same<T, float>::value) {
		if (!fused_capable() || one_pass_ || !fused_ready_ || h_product_ahead_ || prm_.divergence != 0 || !x3_ || !wx3_valid_) return ST_OK;
		GramReduceArgs rgW = gram_args(true, G_, scale_, normalize_next_);
		// (not sampled: the launch belongs to the NEXT iteration -- or to none, when the threshold ends the run here)
		const bool sampled = timing_now_;
		timing_now_ = false;
		const Status s = product_h(Wt_, &rgW, true);
		timing_now_ = sampled;
		if (s != ST_OK) return s;
		h_product_ahead_ = true;
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::iterate(bool compute_error, bool constant_w) {
	const T eps = std::numeric_limits<T>::epsilon();
	timing_now_ = timing_ && (timing_iter_++ % timing_stride_ == 0);
	if (prm_.divergence != 0) return constant_w ? ST_INVALID : iterate_kl(compute_error);
	if (fused_capable() && !constant_w) return iterate_mu64(compute_error);
	if (fused64_capable() && !constant_w) return iterate_fused64(compute_error);
	if (f32w_scale_ != nullptr && fused32w_capable() && !constant_w) return iterate_fused32w(compute_error);
	const int norm_parts = panel_update_parts(RP_, sizeof(T), (int)mpad_);
	h_partials_unneeded_ = !constant_w;              // (the fused iteration: H H^T rides the product where it can)
	const Status hs = h_step_impl(compute_error);
	h_partials_unneeded_ = false;
	if (hs != ST_OK) return hs;

	const bool ls_family = alg_ == ALG_ALS || alg_ == ALG_ACLS || alg_ == ALG_AHCLS;
	int error_terms_n = n_;   // length of the tr(H^T W^T V) term vector

	if (!(constant_w && !compute_error)) {
		const T* Fh = H_;
		bool hht_done = false;
		GramReduceArgs tri_ride = {nullptr, 0, nullptr, nullptr, 0};
		if (tri_) {
			// (riding in the product launch below; on error iterations the trace that reads H H^T then runs behind that launch)
			if (Status s = tri_prepare_h(HHt_, true, !constant_w ? &tri_ride : nullptr)) return s;
			hht_done = true;
		} else if (alg_ == ALG_NSNMF) {
			const T off = (T)prm_.theta / (T)(unsigned)r_;
			const T diag = (T)((1.0 - (T)prm_.theta) + off);
			HIPX(launch_smooth_panel<T>(H_, Hs_, RP_, r_, npad_, off, diag, stream_));
			Fh = Hs_;
		}
		// GDCLS at padded rank 64 on the split-operand path (round 4): H H^T only feeds the W update, which runs AFTER the product against V -- so it is taken from
		// the split image of H by the passenger workgroups of that launch, as the multiplicative update does (gram_image.h): no reduction launch, no partial Gram
		// matrices out of the H update.  (The ALS family inverts H H^T beside the product and needs it before.)
		bool hht_rides = false;
		if constexpr (std::is_same<T, float>::value) {
			hht_rides = !constant_w && alg_ == ALG_GDCLS && x3_ && hx3_valid_ && Fh == H_ && RP_ == 64 && !gram_h_partials_ && passengers_ride(planWx_);
			if (!hht_rides && gram_h_partials_ && Fh == H_) {
				// H H^T from the partial Gram matrices the H update left behind
				HIPX(launch_gram64_from_partials(gramH_part_, (int)(npad_ / 64), HHt_, nullptr, stream_));
				hht_done = true;
			}
		}
		if (!hht_done && !hht_rides) HIPX(launch_gram<T>(Fh, RP_, n_, gram_parts_, gram_part_, HHt_, stream_));
		if (compute_error && !hht_rides && tri_ride.tri_frags == nullptr) {
			const T* wtw = tri_ ? reinterpret_cast<const T*>(Gw_raw_) : G_;      // MU: W^T W of this iteration's H step (rank-256 bf16 path: the unscaled Gram matrix + tri_trace_scale())
			if (alg_ == ALG_NSNMF) {                            // unsmoothed W^T W (AlgorithmNonSmoothNMF.h:201-202)
				if (tri_) wtw = reinterpret_cast<const T*>(Gw_raw_);
				else if (fused_capable()) wtw = G_;             // (rank 64: the H step's passengers took it from the unsmoothed image, pending scale applied)
				else { HIPX(launch_gram<T>(Wt_, RP_, m_, gram_parts_, gram_part_, G2_, stream_)); wtw = G2_; }
			} else if (alg_ != ALG_MU) wtw = G2_;               // LS algorithms: copy saved before the regulariser
			HIPX(launch_trace_small<T>(HHt_, wtw, RP_, r_, psR_, stream_, tri_trace_scale()));
		}
		if (!constant_w) {
			const int S = planW_.splits;
			T offW = 0, diagW = 0;
			if (ls_family) {
				if (alg_ == ALG_ACLS) diagW = (T)prm_.lambdaW;
				else if (alg_ == ALG_AHCLS) {
					const T lam = (T)prm_.lambdaW, alpha = (T)prm_.alphaW;
					T beta = (T)((1 - alpha) * std::sqrt((double)(unsigned)r_) + alpha);
					beta *= beta;
					offW = -lam; diagW = lam * beta - lam;
				}
				// (the trace kernel above has read H H^T; the inverse destroys it) -- beside the product against V
				if (!inverse_rides(planW_)) { if (Status s = normal_inverse_fork(HHt_, offW, diagW)) return s; }
			}
			if (ls_family && inverse_rides(planW_)) {
				if constexpr (std::is_same<T, float>::value) {
					GramReduceArgs rg = {nullptr, 0, nullptr, nullptr, 0};
					rg.inv_a = HHt_; rg.inv_out = Qinv_; rg.inv_offdiag = offW; rg.inv_diag = diagW; rg.inv_r = r_;
					if (Status s = product_w(Fh, &rg, nullptr, x3_ && hx3_valid_ && Fh == H_)) return s;
				}
			} else if (hht_rides) {
				if constexpr (std::is_same<T, float>::value) {
					GramReduceArgs rg = {nullptr, 0, HHt_, nullptr, 0};
					rg.image = Hx3_; rg.image_ks = ksW_;
					if (Status s = product_w(Fh, &rg, nullptr, true)) return s;
					// (the error term's trace reads the finished H H^T: behind the launch that made it)
					if (compute_error) HIPX(launch_trace_small<T>(HHt_, G2_, RP_, r_, psR_, stream_, tri_trace_scale()));
				}
			} else {
				if (Status s = product_w(Fh, tri_ride.tri_frags != nullptr ? &tri_ride : nullptr, nullptr, tri_ || (x3_ && hx3_valid_ && Fh == H_))) return s;
				// (rank 256, H H^T rode in that launch: the error term's trace of H H^T against the unsmoothed W^T W of this iteration's H step, AlgorithmNonSmoothNMF.h:201-202)
				if (compute_error && tri_ride.tri_frags != nullptr) HIPX(launch_trace_small<T>(HHt_, reinterpret_cast<const T*>(Gw_raw_), RP_, r_, psR_, stream_, tri_trace_scale()));
			}
			if (!ls_family) {
				const bool gd_err = alg_ == ALG_GDCLS && compute_error;
				T* wpart = nullptr;
				if constexpr (std::is_same<T, float>::value) { if (gram_from_update()) wpart = gramW_part_; }
				wx3_valid_ = false;
				if (tri_) {
					if (Status s = tri_update_w(slabs_, S, slab_stride_, qx3_holds_hht_ ? nullptr : HHt_)) return s;
				} else {
					HIPX(launch_panel_update<T>(PANEL_MU, Wt_, slabs_, S, slab_stride_, HHt_, RP_, (int)mpad_, eps,
					                            nullptr, m_, sumsq_part_, gd_err ? numW_ : nullptr, stream_, wpart, nullptr, 0, qx3_));
					if (Status s = normalize_w(wpart != nullptr, norm_parts)) return s;
				}
				if (gd_err) {
					// tr(H^T W^T V) as diag((V H^T)^T W) with the UPDATED W (GDCLS :259-264)
					HIPX(launch_row_dot<T>(numW_, Wt_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
					error_terms_n = r_;
				}
			} else {
				if (!inverse_rides(planW_)) { if (Status s = normal_inverse_join()) return s; }
				if (compute_error) HIPX(hipMemcpyAsync(Wold_, Wt_, sizeof(T) * (size_t)RP_ * mpad_, hipMemcpyDeviceToDevice, stream_));
				T* wpart = nullptr;
				if constexpr (std::is_same<T, float>::value) { if (gram_from_update()) wpart = gramW_part_; }
				wx3_valid_ = false;
				HIPX(launch_panel_update<T>(PANEL_LS, Wt_, slabs_, S, slab_stride_, Qinv_, RP_, (int)mpad_, eps,
				                            nullptr, m_, sumsq_part_, compute_error ? numW_ : nullptr, stream_, wpart, nullptr, 0, qx3_));
				if (compute_error) {
					// tr(W_old^T (V H^T)) over r diagonals (ALS :199-205)
					HIPX(launch_row_dot<T>(Wold_, numW_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
					error_terms_n = r_;
				}
				if (Status s = normalize_w(wpart != nullptr, norm_parts)) return s;
			}
		} else if (compute_error && alg_ != ALG_MU && alg_ != ALG_NSNMF) {
			// constant basis vectors, LS algorithms: the reference's trace reads W against itself
			// (ALS/ACLS/AHCLS :199-205 with W never overwritten) or a stale buffer (GDCLS); here:
			// ALS family as the reference, GDCLS the product the formula names.
			if (ls_family) {
				HIPX(launch_row_dot<T>(Wt_, Wt_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
			} else {
				if (Status s = product_w(Fh)) return s;
				HIPX(launch_panel_update<T>(PANEL_SET, numW_, slabs_, planW_.splits, slab_stride_, HHt_, RP_, (int)mpad_, eps,
				                            nullptr, m_, nullptr, nullptr, stream_));
				HIPX(launch_row_dot<T>(numW_, Wt_, RP_, r_, mpad_, psN_, stream_, rowdot_part_));
			}
			error_terms_n = r_;
		}
	}
	if (compute_error) {
		if (Status s = fetch_error_terms(error_terms_n)) return s;
	}
	return ST_OK;
}

// Sparse mode: 0-based triplets (any order) -> CSR and CSC images on the device, the permutation
// between the two value orders, the sorted tr(V^T V) terms and sum(V).  Built on the host with stable
// counting sorts, once per upload (outside the iteration loop).  Duplicate coordinates stay separate
// entries (their contributions add; the densifying path adds them too, k_densify).
template <typename T>
Status Engine<T>::upload_triplets(std::vector<int>& rows, std::vector<int>& cols, std::vector<T>& vals) {
	const long nnz = (long)vals.size();
	if (nnz >= (1l << 31)) return ST_INVALID;
	std::vector<int> csr_ptr(m_ + 1, 0), csc_ptr(n_ + 1, 0), csr_idx(nnz), csc_idx(nnz), from_csr(nnz), order(nnz);
	std::vector<T> csr_val(nnz), csc_val(nnz);
	for (long p = 0; p < nnz; ++p) ++csr_ptr[rows[p] + 1];
	for (int i = 0; i < m_; ++i) csr_ptr[i + 1] += csr_ptr[i];
	{
		std::vector<int> fill(csr_ptr.begin(), csr_ptr.end() - 1);
		for (long p = 0; p < nnz; ++p) order[fill[rows[p]]++] = (int)p;       // stable by row
	}
	// inside a row: ascending column (stable), so that CSR / CSC / COO inputs of one matrix agree bit for bit
	// (a row whose columns already ascend -- what a CSR input usually is -- needs no sort: the check is one pass, the sort allocates and merges per row:
	//  ~0.2 s of config 3's 0.58 s set-up)
	for (int i = 0; i < m_; ++i) {
		const auto lo = order.begin() + csr_ptr[i], hi = order.begin() + csr_ptr[i + 1];
		auto by_col = [&](int x, int y) { return cols[x] < cols[y]; };
		if (!std::is_sorted(lo, hi, by_col)) std::stable_sort(lo, hi, by_col);
	}
	for (long q = 0; q < nnz; ++q) { csr_idx[q] = cols[order[q]]; csr_val[q] = vals[order[q]]; }
	for (long q = 0; q < nnz; ++q) ++csc_ptr[csr_idx[q] + 1];
	for (int j = 0; j < n_; ++j) csc_ptr[j + 1] += csc_ptr[j];
	{
		std::vector<int> fill(csc_ptr.begin(), csc_ptr.end() - 1);
		int row = 0;
		for (long q = 0; q < nnz; ++q) {
			while (q >= csr_ptr[row + 1]) ++row;
			const int dst = fill[csr_idx[q]]++;
			csc_idx[dst] = row; csc_val[dst] = csr_val[q]; from_csr[dst] = (int)q;
		}
	}
	// tr(V^T V) terms per column, accumulated in T like the trace kernel, and sum(V) for the KL divergence
	h_vtv_.assign(n_, T(0));
	sum_v_ = 0;
	for (int j = 0; j < n_; ++j) {
		T s = 0;
		for (int p = csc_ptr[j]; p < csc_ptr[j + 1]; ++p) { s += csc_val[p] * csc_val[p]; sum_v_ += (double)csc_val[p]; }
		h_vtv_[j] = s;
	}
	std::sort(h_vtv_.begin(), h_vtv_.end());

	void** old[] = {(void**)&csr_ptr_, (void**)&csr_idx_, (void**)&csc_ptr_, (void**)&csc_idx_, (void**)&csc_from_csr_, (void**)&csr_val_, (void**)&csc_val_, (void**)&q_, (void**)&q2_};
	for (void** b : old) { if (*b) (void)hipFree(*b); *b = nullptr; }
	const size_t ni = sizeof(int) * (size_t)std::max<long>(nnz, 1), nv = sizeof(T) * (size_t)std::max<long>(nnz, 1);
	HIPX(hipMalloc((void**)&csr_ptr_, sizeof(int) * (m_ + 1)));
	HIPX(hipMalloc((void**)&csc_ptr_, sizeof(int) * (n_ + 1)));
	// (the quotient buffers and the CSR -> CSC permutation only serve round 1's two-pass KL step, a measurement switch)
	const bool two_pass = tuning_env("NMFAMD_KL_TWO_PASS") != nullptr;
	HIPX(hipMalloc((void**)&csr_idx_, ni)); HIPX(hipMalloc((void**)&csc_idx_, ni));
	HIPX(hipMalloc((void**)&csr_val_, nv)); HIPX(hipMalloc((void**)&csc_val_, nv));
	if (two_pass) { HIPX(hipMalloc((void**)&csc_from_csr_, ni)); HIPX(hipMalloc((void**)&q_, nv)); HIPX(hipMalloc((void**)&q2_, nv)); }
	HIPX(hipMemcpyAsync(csr_ptr_, csr_ptr.data(), sizeof(int) * (m_ + 1), hipMemcpyHostToDevice, stream_));
	HIPX(hipMemcpyAsync(csc_ptr_, csc_ptr.data(), sizeof(int) * (n_ + 1), hipMemcpyHostToDevice, stream_));
	if (nnz > 0) {
		HIPX(hipMemcpyAsync(csr_idx_, csr_idx.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, stream_));
		HIPX(hipMemcpyAsync(csc_idx_, csc_idx.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, stream_));
		if (two_pass) HIPX(hipMemcpyAsync(csc_from_csr_, from_csr.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, stream_));
		HIPX(hipMemcpyAsync(csr_val_, csr_val.data(), sizeof(T) * nnz, hipMemcpyHostToDevice, stream_));
		HIPX(hipMemcpyAsync(csc_val_, csc_val.data(), sizeof(T) * nnz, hipMemcpyHostToDevice, stream_));
	}
	nnz_ = nnz;
	if (Status st = setup_kl_blocks()) return st;
	HIPX(hipStreamSynchronize(stream_));
	nnz_ = nnz;
	return ST_OK;
}

// Blocked KL gather (kernels_sparse.hip, k_kl_fused): block counts from the shape and the device, per-(row, block) boundary pointers from the images on the device
template <typename T>
Status Engine<T>::setup_kl_blocks() {
	// KL divergence: cut the gathered factor into blocks that stay in an XCD's L2 (4 MiB; NMFAMD_KL_BLOCK_KB sets the block's bytes,
	// 0 = no blocking) when the whole factor does not: W step gathers rows of H by column index (range n), H step rows of Wt by row index
	// (range m).  A row's entries are sorted by that index, so its entries of block b are one range: only boundaries are needed.
	{
		void** oldb[] = {(void**)&csr_bptr_, (void**)&csc_bptr_, (void**)&kl_part_, (void**)&kl_tpart_};
		for (void** b : oldb) { if (*b) (void)hipFree(*b); *b = nullptr; }
		kl_blocks_w_ = kl_blocks_h_ = 1;
		const char* be = std::getenv("NMFAMD_KL_BLOCK_KB");
		const long block_bytes = be ? std::atol(be) * 1024 : 3584 * 1024;     // (measured at config 3: 2 MiB 1.73 ms / iteration, 3 MiB 1.62, 3.5 MiB 1.58, 4 MiB 1.57, 8 MiB 1.64; unblocked 2.14)
		if (prm_.divergence != 0 && block_bytes > 0) {
			const long rows_per_block = std::max<long>(64, block_bytes / ((long)RP_ * (long)sizeof(T)));
			auto cut = [&](long range) { return (range * RP_ * (long)sizeof(T) > 3l * 1024 * 1024) ? (int)std::min<long>(64, (range + rows_per_block - 1) / rows_per_block) : 1; };
			kl_blocks_w_ = cut(n_); kl_blocks_h_ = cut(m_);
			// five blocks or more: a multiple of eight, so that every XCD gathers from its OWN blocks (k_kl_fused) and fetches 1 / 8 of the factor per launch
			auto by_xcd = [](int b) { return b >= 5 ? ((b + 7) / 8) * 8 : b; };
			kl_blocks_w_ = by_xcd(kl_blocks_w_); kl_blocks_h_ = by_xcd(kl_blocks_h_);
			// every block writes its own partial numerator panel (and k_kl_update reads them all): keep that scratch within an eighth of the DEVICE's memory
			// (64 blocks of a 1M-row factor were 32 GB), halving the block count until it fits.  A function of the shape and of the device only
			// (round 4 asked the FREE memory, so the block count -- and with it the order of the partial sums, i.e. the bits -- depended on what else
			// occupied the device); if the allocation then fails, the unblocked gather runs (below).  geometry() reports the counts that ran.
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)1 << 62; }
			auto fit = [&](int b, long panel_rows) { while (b > 1 && (double)b * RP_ * (double)panel_rows * sizeof(T) > 0.125 * (double)total_b) b = b > 8 ? ((b / 2 + 7) / 8) * 8 : b / 2; return b; };
			kl_blocks_w_ = fit(kl_blocks_w_, mpad_); kl_blocks_h_ = fit(kl_blocks_h_, npad_);
			auto boundaries = [&](const int* ptr, const int* idx, int rows, long range, int blocks, int** dev) -> hipError_t {
				if (blocks <= 1) return hipSuccess;
				hipError_t e = hipMalloc((void**)dev, sizeof(int) * (size_t)rows * (blocks + 1));
				if (e != hipSuccess) return e;
				return launch_sp_boundaries(ptr, idx, rows, range, blocks, *dev, stream_);      // (a row's indices ascend: block b of the row is one range)
			};
			HIPX(boundaries(csr_ptr_, csr_idx_, m_, n_, kl_blocks_w_, &csr_bptr_));
			HIPX(boundaries(csc_ptr_, csc_idx_, n_, m_, kl_blocks_h_, &csc_bptr_));
			const long pe = std::max<long>(kl_blocks_w_ > 1 ? (long)kl_blocks_w_ * RP_ * mpad_ : 0, kl_blocks_h_ > 1 ? (long)kl_blocks_h_ * RP_ * npad_ : 0);
			// (no room for the partial panels after all: the unblocked gather needs none)
			bool ok = pe == 0 || hipMalloc((void**)&kl_part_, sizeof(T) * (size_t)pe) == hipSuccess;
			if (ok && kl_blocks_w_ > 1) ok = hipMalloc((void**)&kl_tpart_, sizeof(T) * 2 * (size_t)kl_blocks_w_ * mpad_) == hipSuccess;
			if (!ok) {
				(void)hipGetLastError();
				for (void** b : oldb) { if (*b) (void)hipFree(*b); *b = nullptr; }
				kl_blocks_w_ = kl_blocks_h_ = 1;
			}
		}
	}
	return ST_OK;
}

// KL-divergence multiplicative update (Lee & Seung, "Algorithms for Non-negative Matrix
// Factorization", NIPS 2001 -- the paper the reference's README cites; the reference itself only has
// the Frobenius form), in the skeleton of the reference's MU iteration: H step, W step, column
// normalisation of W, error terms referring to (W_{k-1}, H_k):
//   Q = V ./ (W H + eps) on the stored entries (SDDMM);  H .*= (W^T Q) ./ (colsum(W) + eps)
//   Q = V ./ (W H + eps) again;                          W .*= (Q H^T) ./ (rowsum(H) + eps);  normalise
template <typename T>
Status Engine<T>::iterate_kl(bool compute_error) {
	if (!sparse_ || nnz_ < 0) return ST_INVALID;
	const T eps = std::numeric_limits<T>::epsilon();
	const int norm_parts = (int)(mpad_ / 128);
	const bool two_pass = tuning_env("NMFAMD_KL_TWO_PASS") != nullptr;      // A/B: round 1's SDDMM + permute + SpMM per half-step
	// Round 6: W is left UNNORMALISED by its update, with the column scale d = 1 / sqrt(sum of squares) as a pending factor (kl_scale_, from the update's partial
	// sums, by the k_kl_sums launch that turns them into the column sums anyway): the pass over the 51 MB panel that normalised it (k_compact_partials +
	// k_normalize_panel_v2: 25 us at config 3) is gone.  W = Wt D enters the two half-steps through the row each fused kernel loads ONCE per output row
	// (a_scale), the H update's numerator is D (Wt^T Q), the W update reads its old rows as Wt D.  materialize_w() folds the scale in for everybody else.
	const bool keep_pending = !two_pass && std::getenv("NMFAMD_NO_FUSED_MU") == nullptr;
	if (!keep_pending) { if (Status st = materialize_w()) return st; }
	const T* dsc = kl_scale_pending_ ? kl_scale_ : nullptr;
	// H step (no error terms here: they refer to the pair (W_{k-1}, H_k) of the second evaluation).  One pass over the CSC
	// image: quotient and numerator W^T Q together, one gathered row of W per stored entry (kernels_sparse.hip, k_kl_fused).
	record_begin();
	if (two_pass) {
		HIPX(launch_sddmm_quotient<T>(csr_ptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, q_, (T*)nullptr, (T*)nullptr, m_, stream_));
		HIPX(launch_permute<T>(q_, csc_from_csr_, q2_, nnz_, stream_));
		HIPX(launch_spmm_rows<T>(csc_ptr_, csc_idx_, q2_, Wt_, RP_, slabs_, n_, (int)npad_, stream_));
	} else {
		if (kl_blocks_h_ > 1) HIPX(launch_kl_fused<T>(csc_bptr_, csc_idx_, csc_val_, H_, Wt_, RP_, eps, kl_part_, (T*)nullptr, (T*)nullptr, n_, (int)npad_, stream_, kl_blocks_h_, (long)RP_ * npad_, dsc));
		else HIPX(launch_kl_fused<T>(csc_ptr_, csc_idx_, csc_val_, H_, Wt_, RP_, eps, slabs_, (T*)nullptr, (T*)nullptr, n_, (int)npad_, stream_, 1, 0, dsc));
	}
	record_end();
	// the column sums of W: from the partial sums its last update left (round 4), or by a pass over the panel when W was set from outside
	if (!kl_sw_ready_) HIPX(launch_panel_rowsum<T>(Wt_, RP_, (int)mpad_, rowsum_part_, sW_, stream_));
	// ... and the update of H leaves the partial row sums of the NEW H (the W step's denominators) in rowsum_part_: one small launch instead of a pass over H
	if (!two_pass && kl_blocks_h_ > 1) HIPX(launch_kl_update<T>(H_, kl_part_, sW_, RP_, (int)npad_, eps, nullptr, stream_, kl_blocks_h_, (long)RP_ * npad_, rowsum_part_, dsc));
	else HIPX(launch_kl_update<T>(H_, slabs_, sW_, RP_, (int)npad_, eps, nullptr, stream_, 1, 0, rowsum_part_, dsc));
	HIPX(launch_kl_sums<T>(rowsum_part_, nullptr, (int)(npad_ / 128), RP_, sH_, stream_));
	// W step (the quotient is re-evaluated with the new H), over the CSR image; per-row error terms on error iterations only
	record_begin(1);
	if (two_pass) {
		HIPX(launch_sddmm_quotient<T>(csr_ptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, q_, compute_error ? t_vwh_ : (T*)nullptr, compute_error ? t_kl_ : (T*)nullptr, m_, stream_));
		HIPX(launch_spmm_rows<T>(csr_ptr_, csr_idx_, q_, H_, RP_, slabs_, m_, (int)mpad_, stream_));
	} else {
		if (kl_blocks_w_ > 1) {
			HIPX(launch_kl_fused<T>(csr_bptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, kl_part_, compute_error ? kl_tpart_ : (T*)nullptr,
			                        compute_error ? kl_tpart_ + (long)kl_blocks_w_ * mpad_ : (T*)nullptr, m_, (int)mpad_, stream_, kl_blocks_w_, (long)RP_ * mpad_, dsc));
			if (compute_error) {
				// the blocks' parts of the per-row error terms, block order
				HIPX(launch_reduce_partials<T>(kl_tpart_, kl_blocks_w_, mpad_, t_vwh_, m_, stream_));
				HIPX(launch_reduce_partials<T>(kl_tpart_ + (long)kl_blocks_w_ * mpad_, kl_blocks_w_, mpad_, t_kl_, m_, stream_));
			}
		} else
		HIPX(launch_kl_fused<T>(csr_ptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, slabs_, compute_error ? t_vwh_ : (T*)nullptr, compute_error ? t_kl_ : (T*)nullptr,
		                        m_, (int)mpad_, stream_, 1, 0, dsc));
	}
	record_end();
	if (compute_error) {
		// Frobenius error by the reference's trace formula with (W_{k-1}, H_k); KL divergence next to it
		HIPX(launch_gram<T>(Wt_, RP_, m_, gram_parts_, gram_part_, G_, stream_));
		HIPX(launch_gram<T>(H_, RP_, n_, gram_parts_, gram_part_, HHt_, stream_));
		HIPX(launch_trace_small<T>(HHt_, G_, RP_, r_, psR_, stream_, nullptr, dsc));      // (G_ is the Gram matrix of the panel as it lies: D G D = W^T W of the normalised W)
		// the terms travel stream-ordered into pinned memory; the sorted host summation (100 000 per-row terms at config 3:
		// milliseconds) runs when the error is read -- the iteration loop does not wait for the GPU here
		finalize_error(false);
		T* p = pin_kl_;
		HIPX(hipMemcpyAsync(p, t_vwh_, sizeof(T) * m_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + m_, t_kl_, sizeof(T) * m_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_, sW_, sizeof(T) * RP_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_ + RP_, sH_, sizeof(T) * RP_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_ + 2 * RP_, psR_, sizeof(T) * r_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipEventRecord(err_event_, stream_));
		kl_pending_ = true;
	}
	if (!two_pass && kl_blocks_w_ > 1) HIPX(launch_kl_update<T>(Wt_, kl_part_, sH_, RP_, (int)mpad_, eps, sumsq_part_, stream_, kl_blocks_w_, (long)RP_ * mpad_, rowsum_part_, dsc));
	else HIPX(launch_kl_update<T>(Wt_, slabs_, sH_, RP_, (int)mpad_, eps, sumsq_part_, stream_, 1, 0, rowsum_part_, dsc));
	// colsum of the NORMALISED W = (sum of the new column) / (its norm), both from the update's per-workgroup partials (norm_parts = mpad / 128 of each); and the
	// pending scale itself
	HIPX(launch_kl_sums<T>(rowsum_part_, sumsq_part_, norm_parts, RP_, sW_, stream_, keep_pending ? kl_scale_ : nullptr));
	kl_sw_ready_ = true;
	if (keep_pending) kl_scale_pending_ = true;
	else HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, norm_parts, stream_));
	return ST_OK;
}

// ---- the KL update in the three phases of the column-sharded form ---------------------------------------------------------------------
// Rank g holds the CSC / CSR images of V(:, J_g), H(:, J_g) and a replica of W.  The H half-step is local.  The W half-step's numerator
// sum_j Q(i, j) H(:, j) and the row sums of H are sums over ALL columns: every rank contributes its columns' part through the exchange buffer
// (layout: Engine::exchange_count), as do the per-row error terms; the W update and its normalisation then run replicated.
template <typename T>
Status Engine<T>::kl_h_step() {
	if (!sparse_ || nnz_ < 0) return ST_INVALID;
	if (Status st = materialize_w()) return st;      // (the sharded form works on the normalised panel)
	const T eps = std::numeric_limits<T>::epsilon();
	record_begin();
	if (kl_blocks_h_ > 1) HIPX(launch_kl_fused<T>(csc_bptr_, csc_idx_, csc_val_, H_, Wt_, RP_, eps, kl_part_, (T*)nullptr, (T*)nullptr, n_, (int)npad_, stream_, kl_blocks_h_, (long)RP_ * npad_));
	else HIPX(launch_kl_fused<T>(csc_ptr_, csc_idx_, csc_val_, H_, Wt_, RP_, eps, slabs_, (T*)nullptr, (T*)nullptr, n_, (int)npad_, stream_));
	record_end();
	HIPX(launch_panel_rowsum<T>(Wt_, RP_, (int)mpad_, rowsum_part_, sW_, stream_));
	if (kl_blocks_h_ > 1) HIPX(launch_kl_update<T>(H_, kl_part_, sW_, RP_, (int)npad_, eps, nullptr, stream_, kl_blocks_h_, (long)RP_ * npad_));
	else HIPX(launch_kl_update<T>(H_, slabs_, sW_, RP_, (int)npad_, eps, nullptr, stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::kl_w_products(T* exchange, bool compute_error) {
	if (!sparse_ || nnz_ < 0) return ST_INVALID;
	const T eps = std::numeric_limits<T>::epsilon();
	T* num = exchange;
	T* hht = num + (long)RP_ * mpad_;
	T* sh = hht + (long)RP_ * RP_;
	T* tv = sh + RP_;
	T* tk = tv + mpad_;
	record_begin(1);
	if (kl_blocks_w_ > 1) {
		HIPX(launch_kl_fused<T>(csr_bptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, kl_part_, compute_error ? kl_tpart_ : (T*)nullptr,
		                        compute_error ? kl_tpart_ + (long)kl_blocks_w_ * mpad_ : (T*)nullptr, m_, (int)mpad_, stream_, kl_blocks_w_, (long)RP_ * mpad_));
		// the blocks' partial numerator panels (and error terms), block order, before the sum over ranks
		HIPX(launch_reduce_partials<T>(kl_part_, kl_blocks_w_, (long)RP_ * mpad_, num, (long)RP_ * mpad_, stream_));
		if (compute_error) {
			HIPX(launch_reduce_partials<T>(kl_tpart_, kl_blocks_w_, mpad_, tv, m_, stream_));
			HIPX(launch_reduce_partials<T>(kl_tpart_ + (long)kl_blocks_w_ * mpad_, kl_blocks_w_, mpad_, tk, m_, stream_));
		}
	} else {
		HIPX(launch_kl_fused<T>(csr_ptr_, csr_idx_, csr_val_, Wt_, H_, RP_, eps, num, compute_error ? tv : (T*)nullptr, compute_error ? tk : (T*)nullptr,
		                        m_, (int)mpad_, stream_));
	}
	record_end();
	HIPX(launch_panel_rowsum<T>(H_, RP_, (int)npad_, rowsum_part_, sh, stream_));
	if (compute_error) HIPX(launch_gram<T>(H_, RP_, n_, gram_parts_, gram_part_, hht, stream_));
	else {
		// the regions nobody reads on this iteration still ride the all-reduce: keep them finite (left alone they grow by the rank count per iteration)
		HIPX(hipMemsetAsync(hht, 0, sizeof(T) * (size_t)RP_ * RP_, stream_));
		HIPX(hipMemsetAsync(tv, 0, sizeof(T) * 2 * (size_t)mpad_, stream_));
	}
	return ST_OK;
}

template <typename T>
Status Engine<T>::kl_w_finish(const T* exchange, bool compute_error) {
	if (!sparse_ || nnz_ < 0) return ST_INVALID;
	// the error terms in the exchange buffer were (or were not) produced by w_products() under the flag h_step() latched: a caller asking for the
	// other thing here would read terms that are not there
	if (compute_error != kl_err_iter_) { last_error_ = "kl: w_finish(compute_error) differs from the h_step(compute_error) of the same iteration"; return ST_INVALID; }
	kl_sw_ready_ = false;
	const T eps = std::numeric_limits<T>::epsilon();
	const int norm_parts = (int)(mpad_ / 128);
	const T* num = exchange;
	const T* hht = num + (long)RP_ * mpad_;
	const T* sh = hht + (long)RP_ * RP_;
	const T* tv = sh + RP_;
	const T* tk = tv + mpad_;
	if (compute_error) {
		// as iterate_kl: Frobenius error by the trace formula with (W_{k-1}, H_k), KL divergence next to it -- from the reduced terms
		HIPX(launch_gram<T>(Wt_, RP_, m_, gram_parts_, gram_part_, G_, stream_));
		HIPX(launch_trace_small<T>(hht, G_, RP_, r_, psR_, stream_));
		finalize_error(false);
		T* p = pin_kl_;
		HIPX(hipMemcpyAsync(p, tv, sizeof(T) * m_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + m_, tk, sizeof(T) * m_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_, sW_, sizeof(T) * RP_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_ + RP_, sh, sizeof(T) * RP_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipMemcpyAsync(p + 2 * (size_t)m_ + 2 * RP_, psR_, sizeof(T) * r_, hipMemcpyDeviceToHost, stream_));
		HIPX(hipEventRecord(err_event_, stream_));
		kl_pending_ = true;
	}
	HIPX(launch_kl_update<T>(Wt_, num, sh, RP_, (int)mpad_, eps, sumsq_part_, stream_));
	HIPX(launch_normalize_panel<T>(Wt_, RP_, (int)mpad_, sumsq_part_, norm_parts, stream_));
	return ST_OK;
}

template <typename T>
Status Engine<T>::debug_read(int which, T* out, long count) {
	const T* src = nullptr; long avail = 0;
	switch (which) {
	case 0: if (Status st = materialize_w()) return st; src = Wt_; avail = (long)RP_ * mpad_; break;
	case 1: src = H_; avail = (long)RP_ * npad_; break;
	case 2: src = G_; avail = (long)RP_ * RP_; break;
	case 3: src = HHt_; avail = (long)RP_ * RP_; break;
	case 4: src = slabs_; avail = slab_stride_ * std::max(planH_.splits, planW_.splits); break;
	case 5: src = Qinv_; avail = (long)RP_ * RP_; break;
	case 8: case 9: case 10: case 11: {
		// rank-256 intermediates, read as fp32 words (diagnosis of runs that share a device: tools/shared_device_diff.py):
		// 8 = W^T W as the last Gram reduction left it, 9 = the staged column sums of squares, 10 / 11 = the bf16 fragments of W / of H
		if (!tri_ || sizeof(T) != 4) return ST_INVALID;
		const void* p = which == 8 ? (const void*)Gw_raw_ : which == 9 ? (const void*)colsq_ : which == 10 ? (const void*)Wtb_ : (const void*)Hb_;
		avail = which == 8 ? (long)RP_ * RP_ : which == 9 ? (long)RP_ : which == 10 ? (long)RP_ * mpad_ / 2 : (long)RP_ * npad_ / 2;
		if (!p || count > avail) return ST_INVALID;
		HIPX(hipMemcpyAsync(out, p, sizeof(T) * count, hipMemcpyDeviceToHost, stream_));
		HIPX(hipStreamSynchronize(stream_));
		return ST_OK;
	}
	case 12: case 13: case 14: case 15: case 16: case 17: case 18: case 19: {
		// the sparse images (tests/test_gpu_sparse_setup.py compares the device-built ones with the host-built ones): 12 / 13 = CSR / CSC values; 14 .. 19 = CSR
		// pointers, CSR column indices, CSC pointers, CSC row indices, blocked CSR / CSC boundary pointers as raw 32-bit words (fp32 engines: count = number of ints)
		if (!sparse_) return ST_INVALID;
		if (which >= 14 && sizeof(T) != 4) return ST_INVALID;
		const long bw = kl_blocks_w_ > 1 ? (long)m_ * (kl_blocks_w_ + 1) : 0, bh = kl_blocks_h_ > 1 ? (long)n_ * (kl_blocks_h_ + 1) : 0;
		const void* ptrs[] = {csr_val_, csc_val_, csr_ptr_, csr_idx_, csc_ptr_, csc_idx_, csr_bptr_, csc_bptr_};
		const long lens[] = {nnz_, nnz_, (long)m_ + 1, nnz_, (long)n_ + 1, nnz_, bw, bh};
		const void* sp = ptrs[which - 12];
		if (sp == nullptr || count > lens[which - 12]) return ST_INVALID;
		HIPX(hipMemcpyAsync(out, sp, sizeof(T) * count, hipMemcpyDeviceToHost, stream_));
		HIPX(hipStreamSynchronize(stream_));
		return ST_OK;
	}
	case 6: case 7: {
		if (sparse_ || bf16_ || (one_image_ && which == 7)) return ST_INVALID;   // no fp32 dense image in sparse / bf16 mode; no V^T image
		// V (ld mpad_) / Vt (ld npad_) as column-major images; the MFMA path keeps them x-tiled
		const bool vt = which == 7;
		const T* img = vt ? Vt_ : V_;
		avail = mpad_ * npad_;
		if (count > avail) return ST_INVALID;
		if (!tiled_) { src = img; break; }
		T* tmp = nullptr;
		HIPX(hipMalloc((void**)&tmp, sizeof(T) * (size_t)avail));
		hipError_t e = hipMemsetAsync(tmp, 0, sizeof(T) * (size_t)avail, stream_);
		if (e == hipSuccess) e = vt ? launch_tile<T>(img, npad_, n_, m_, tmp, strideVt_, planH_.th, true, stream_) : launch_tile<T>(img, mpad_, m_, n_, tmp, strideV_, img_th_, true, stream_);
		if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, sizeof(T) * count, hipMemcpyDeviceToHost, stream_);
		if (e == hipSuccess) e = hipStreamSynchronize(stream_);
		(void)hipFree(tmp);
		return e == hipSuccess ? ST_OK : hip_fail(e, "debug_read(V)");
	}
	default: return ST_INVALID;
	}
	if (count > avail) return ST_INVALID;
	HIPX(hipMemcpyAsync(out, src, sizeof(T) * count, hipMemcpyDeviceToHost, stream_));
	HIPX(hipStreamSynchronize(stream_));
	return ST_OK;
}

template class Engine<float>;
template class Engine<double>;

} // namespace nmfamd
