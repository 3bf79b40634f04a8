// engine.h -- device-resident state and per-iteration orchestration of one factorisation
// (internal header).  This is the MI355X-native replacement of the reference's L2/L1 layers:
// IAlgorithm + the five Algorithm*.h classes (source/nmf/Algorithm.h:38-77) and DeviceMatrix
// (source/common/Matrix.h:446-645).  Everything stays in HBM between upload and download;
// the host sees n + r partial sums on error iterations, exactly like the reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "kernels.h"

namespace nmfamd {

enum Algorithm { ALG_MU = 0, ALG_GDCLS = 1, ALG_ALS = 2, ALG_ACLS = 3, ALG_AHCLS = 4, ALG_NSNMF = 5 };

struct AlgorithmParams {
	double lambda = 0, lambdaW = 0, lambdaH = 0, alphaW = 0, alphaH = 0, theta = 0;
	// extensions without a reference counterpart (selected by new Parameter names, see abi.cpp):
	double divergence = 0;      // 0: Frobenius objective, 1: generalised KL divergence (multiplicative update only)
	double sparse_compute = 0;  // 1: keep V as CSR + CSC in HBM and multiply by SpMM instead of densifying
	double precision = 0;       // 1: bf16 MFMA operands (V, W, H rounded to bf16 inside the two big products), fp32 everywhere else
};

// Status codes shared with nmfgpu_amd.h (NMFAMD_*).
// ST_VALUE_RANGE: V holds values the split-operand product is not exact for (infinities, NaN, |v| > 2^126, 0 < |v| < 2^-100):
// the caller recreates the engine with precision = -1 (native fp32 MFMA instructions) -- nmfgpu::compute does that by itself
enum Status { ST_OK = 0, ST_INVALID = 1, ST_NO_DEVICE_MEMORY = 2, ST_NO_HOST_MEMORY = 3, ST_HIP_ERROR = 4, ST_NO_DEVICE = 5, ST_VALUE_RANGE = 6 };

// Padded rank of the factor panels: 64, or above that a multiple of 128 in fp32 (the 128-column forms of the split-operand product, the wide fp32 update and Gram
// kernels) and of 64 in fp64 (round 5: every fp64 kernel works in 64-column units -- the reference example's r = 158 ran as 256, now 192: 0.56 of the r x r work
// and 0.75 of the product's)
inline int padded_rank(int r, size_t elem_bytes = 4) { return r <= 64 ? 64 : (elem_bytes == 8 ? ((r + 63) / 64) * 64 : ((r + 127) / 128) * 128); }
inline long pad128(long v) { return ((v + 127) / 128) * 128; }

template <typename T>
class Engine {
public:
	Engine(int m, int n, int r, int algorithm, const AlgorithmParams& params);
	~Engine();

	Status allocate();
	// Row-block form of the sharded W step (sharded.cpp): the padded row count becomes a multiple of 128 * blocks, so that
	// the Wt panel splits into `blocks` equal row blocks.  Call before allocate().
	void set_row_blocks(int blocks) { row_blocks_ = blocks > 1 ? blocks : 1; }
	// the sharded three-phase API driven by a team of ONE rank: the exchange buffer goes from w_products() to w_finish() as it is
	void set_sole_rank(bool sole) { sole_rank_ = sole; }
	// The one-pass iteration (kernels_onepass.hip) claims every CU of the device for one persistent launch: engines that run
	// beside others on one device (rank threads of a team) opt out.  Call before allocate().
	void set_one_pass(bool allow) { one_pass_allowed_ = allow; }
	// 0: not in use, 1: in use, 2: was in use and gave up (a launch could not form its groups; the two-pass iteration took over)
	int one_pass_state() const { return one_pass_ ? 1 : (one_pass_gave_up_ ? 2 : 0); }
	void set_stream(hipStream_t s) { stream_ = s; }
	hipStream_t stream() const { return stream_; }

	// V: host, column-major / sparse.  Builds V, Vt and the sorted tr(V^T V) vector.
	Status upload_dense(const T* V, long ld);
	Status upload_sparse(int format, const T* values, const int* a, const int* b, long nnz, int base);

	// W: host m x r (ld), H: host r x n (ld).  Either pointer may be null (leave as is).
	Status set_factors(const T* W, long ldw, const T* H, long ldh);
	Status get_factors(T* W, long ldw, T* H, long ldh);   // nsNMF returns W S, like the reference
	// h_first_column: global index of this engine's first column (column shards draw their part of ONE stream)
	Status randomize_factors(unsigned seed, bool w, bool h, long h_first_column = 0);

	// The W^T V launch (and its Gram passengers) of the NEXT iteration, enqueued ahead of it: it writes the split-K slabs, G and the column scale -- scratch the
	// next iterate() would overwrite anyway -- so a caller that is about to block on this iteration's error value (nmfgpu::compute: the threshold test decides
	// whether there is a next iteration) keeps the device busy meanwhile.  Rank-64 fused path only; a no-op elsewhere.  The next iterate() skips the launch;
	// set_factors / randomize / upload in between make it void.
	Status begin_next_iteration();
	// One iteration.  With compute_error the frobenius()/rmsd() values are refreshed (host sync).
	Status iterate(bool compute_error, bool constant_w);

	// Split form for column-sharded multi-GPU runs (exchange = device buffer of exchange_count()
	// elements: the local (V H^T)^T panel followed by the local H H^T):
	//   h_step -> w_products(exchange) -> [all-reduce exchange across ranks] -> w_finish(exchange)
	Status h_step(bool compute_error);
	Status w_products(T* exchange);
	Status w_finish(const T* exchange, bool compute_error);
	// ... and without a reduction in between: exchanges[p] = rank p's exchange buffer as its w_products() left it, readable from this device (count ranks, rank
	// order).  Available where direct_w_finish() says so (rank-64 multiplicative update on the split-operand path).
	Status w_finish_peers(const T* const* exchanges, int count, bool compute_error);
	bool direct_w_finish() const { return fused_capable() && gram_image_ && prm_.divergence == 0; }
	// KL update (sparse V): the exchange also carries the row sums of the local H and, on error iterations, the per-row error terms:
	//   [ numerator panel RP x mpad | H_g H_g^T RP x RP | rowsum(H_g) RP | tr terms mpad | KL terms mpad ]
	long exchange_count() const { return (long)RP_ * mpad_ + (long)RP_ * RP_ + (prm_.divergence != 0 ? (long)RP_ + 2 * mpad_ : 0); }
	bool is_kl() const { return prm_.divergence != 0; }
	// sharded runs: the error terms refer to the whole matrix (sorted tr(V^T V) terms of ALL columns, sum of ALL entries, total column count)
	void set_error_globals(const std::vector<T>& vtv_sorted_all, double sum_v_all, long total_columns) { h_vtv_ = vtv_sorted_all; sum_v_ = sum_v_all; err_total_columns_ = total_columns; }
	double sum_v() const { return sum_v_; }
	// Row-block form of w_finish (reduce-scatter by row blocks of W -> every GPU updates its rows -> all-reduce
	// of the r column-norm partials -> all-gather of the normalised rows).  num_rows: the reduced (V H^T)^T rows
	// [row0, row0 + rows) in panel layout; hht: the reduced H H^T.  w_update_rows leaves the r partial sums of squares of
	// the new rows in colsq (device, RP elements); after their all-reduce w_normalize_rows scales the block, and once the
	// blocks of every rank have been gathered into w_panel(), w_rows_replaced() drops what was derived from the old W.
	Status w_update_rows(const T* num_rows, const T* hht, long row0, long rows, bool compute_error, T* colsq);
	Status w_normalize_rows(long row0, long rows, T* colsq);
	void w_rows_replaced() { kl_sw_ready_ = false; kl_scale_pending_ = false; fused_ready_ = false; w_pending_ = false; f32w_pending_ = false; f64_pending_ = false; f64_product_ahead_ = false; h_product_ahead_ = false; gram_w_ready_ = false; wx3_valid_ = false; tri_gw_ready_ = false; qx3_holds_g_ = false; tri_scale_pending_ = false; tri_scale_from_gram_ = false; if (!tri_rows_cover_) wtb_valid_ = false; }
	T* w_panel() { return Wt_; }
	int kl_blocks(bool w_step) const { return prm_.divergence != 0 ? (w_step ? kl_blocks_w_ : kl_blocks_h_) : 0; }
	int gram_k_slices() const { return gram_spread_ ? GRAM_REDUCE_BLOCKS : gram_ksplit_; }      // (16: the spread form)
	bool w_col_split() const { return w_col_split_; }
	int fused_launches() const {
		if ((fused_capable() && !one_pass_ && gram_image_) || (f64_partial_ != nullptr && fused64_capable())) return 4;
		return f32w_scale_ != nullptr && fused32w_capable() ? 8 - (f32w_ride_h_ > 0 ? 1 : 0) - (f32w_ride_w_ > 0 ? 1 : 0) : 0;
	}
	int gram_ride_slices(bool w_side) const { return f64_partial_ != nullptr ? (w_side ? f64_slices_w_ : f64_slices_h_) : (f32w_scale_ != nullptr ? (w_side ? f32w_ride_w_ : f32w_ride_h_) : 0); }
	// Row-block form at padded rank 256 with bf16 operands (config 4): between two W updates the OTHER ranks read only the bf16 fragments of a rank's rows (the next
	// W^T V's operand and the Gram matrix are made from them) -- so the all-gather carries the fragments w_normalize_rows() left for this rank's rows (RP / 2 four-byte
	// words per row: 25.6 MB at config 4 instead of 51.2 MB of fp32 rows) and the fp32 rows of the other ranks stay STALE in w_panel() until somebody needs them
	// (get_factors, a fresh set of fragments): then the hook -- the sharded run's all-gather of the fp32 row blocks, a COLLECTIVE -- runs first.
	bool w_fragment_exchange() const { return tri_; }
	void* w_fragments() { return Wtb_; }
	long w_fragment_words_per_row() const { return RP_ / 2; }
	// the fragments of every rank's rows have been gathered into w_fragments(); stale: the fp32 rows of the other ranks were not
	void w_fragments_gathered(bool stale) {
		kl_sw_ready_ = false; kl_scale_pending_ = false; fused_ready_ = false; w_pending_ = false; f32w_pending_ = false; f64_pending_ = false; f64_product_ahead_ = false; h_product_ahead_ = false; gram_w_ready_ = false; wx3_valid_ = false; tri_gw_ready_ = false; qx3_holds_g_ = false;
		tri_scale_pending_ = false; tri_scale_from_gram_ = false; wtb_valid_ = true; w_rows_stale_ = stale;
	}
	void set_w_gather_hook(std::function<Status()> hook) { w_gather_hook_ = std::move(hook); }
	void w_rows_gathered() { w_rows_stale_ = false; }
	bool w_rows_stale() const { return w_rows_stale_; }
	Status materialize() { return materialize_w(); }
	// error terms of the last error iteration (host copies): n_local per-column terms and r terms
	const std::vector<T>& terms_htwtv() { finalize_error(false); return h_psN_; }
	const std::vector<T>& terms_hhtwtw() { finalize_error(false); return h_psR_; }
	const std::vector<T>& terms_vtv_sorted() const { return h_vtv_; }
	long error_terms_to_device(T* dst, long capacity);   // [psN (last count) | psR (r)], D2D on the stream
	// sharded runs: the terms are fetched with error_terms_to_device() only -- the engine's own copy to the host (and its event) is skipped
	void set_error_terms_stay_on_device(bool stay) { error_terms_stay_ = stay; }
	void resolve_error(const std::vector<T>& vtv_sorted, std::vector<T> htwtv, std::vector<T> hhtwtw, long total_elements);      // (sorts copies: terms_htwtv() keeps the column order)

	// Error of the most recent error iteration.  The n + r partial sums travel to the host
	// asynchronously; the first reader waits for them and does the sorted summation, so a caller
	// that does not look at the error every time (the benchmark loop) never stalls the stream.
	double kl_divergence() { finalize_error(true); return kl_; }
	bool sparse_mode() const { return sparse_; }
	bool sparse_setup_on_device() const { return sparse_setup_on_device_; }
	long nnz() const { return nnz_; }
	double frobenius() { finalize_error(true); return frob_; }
	double frobenius_squared() { finalize_error(true); return frob2_; }      // the resolved sum before the root
	double rmsd() { finalize_error(true); return rmsd_; }

	// Timing of the dominant kernel (the factor product): when enabled, every launch is bracketed
	// by HIP events on the engine's stream; dominant_stats reads them back (host sync).
	// stride: time the launches of every stride-th iteration only (1 = all)
	void enable_kernel_timing(bool on, int stride = 1) { timing_ = on; timing_stride_ = stride > 0 ? stride : 1; timing_iter_ = 0; }
	// kind_ms / kind_launches (optional, two entries each): the same split by product -- [0] the H-side product (W^T V, or the KL H half-step's gather), [1] the W side
	void dominant_stats(double* total_ms, long* launches, double* pair_overhead_ms = nullptr, double* kind_ms = nullptr, long* kind_launches = nullptr);

	int m() const { return m_; }
	int n() const { return n_; }
	int r() const { return r_; }
	int rp() const { return RP_; }
	// GDCLS and the ALS family evaluate tr(H^T W^T V) as r terms (one per factor row, from the reduced sums: identical on every rank of a
	// column-sharded run); the multiplicative algorithms as one term per column of V
	bool error_terms_per_factor_row() const { return alg_ != ALG_MU && alg_ != ALG_NSNMF; }
	long mpad() const { return mpad_; }
	long npad() const { return npad_; }
	int slabs_h() const { return planH_.splits; }
	int slabs_w() const { return planW_.splits; }
	// which kernel runs the two big products: 0 fp32 MFMA, 1 bf16-rounded operands, 2 fp32 by exact 3 x bf16 splitting,
	// 3 fp64 MFMA, 4 VALU fallback kernel (NMFAMD_FORCE_VALU), 5 sparse (SpMM)
	int resident_images() const { return sparse_ ? 0 : (one_image_ ? 1 : 2); }
	int product_kernel() const { return sparse_ ? 5 : bf16_ ? 1 : x3_ ? 2 : !tiled_ ? 4 : (sizeof(T) == 8 ? 3 : 0); }
	const char* last_error() const { return last_error_; }

	// test access to device intermediates (panel layout, host copies)
	Status debug_read(int which, T* out, long count);

private:
	Status hip_fail(hipError_t e, const char* what);
	Status h_step_impl(bool compute_error);
	// prepacked: the split (x3) image of F is already in Wx3_ / Hx3_ (emitted by the update kernel that wrote F)
	Status product_h(const T* F, const GramReduceArgs* rg = nullptr, bool prepacked = false);   // slabs_ <- partials of F V   (r x n)
	Status product_w(const T* F, const GramReduceArgs* rg = nullptr, T* single_slab_out = nullptr, bool prepacked = false);   // slabs_ (or the caller's panel when there is one K slice) <- partials of (V F^T)^T (r x m)
	bool fused_capable() const;                      // fp32, padded rank 64, MU (and nsNMF on the split-operand products)
	bool fused32w_capable() const;                   // fp32, split-operand products, MU / nsNMF, padded ranks 128 ... 512: the eight-launch iteration of iterate_fused32w
	Status iterate_fused32w(bool compute_error);     // (Gram slices + reduce) / product / update, twice; W carried unnormalised with a pending column scale, no pack / smooth / normalise launches
	bool fused64_capable() const;                    // fp64, MU / nsNMF, any padded rank up to 512: the four-launch iteration of iterate_fused64
	Status iterate_fused64(bool compute_error);      // product (+ Gram passengers) / update / product (+ Gram passengers) / update, W carried unnormalised with a pending column scale
	bool gram_from_update() const;                   // GDCLS / ALS family at fp32, padded rank 64: Gram matrices from the update kernel's partials
	Status iterate_mu64(bool compute_error);         // the four-launch iteration of kernels_mu64.hip
	Status iterate_onepass(bool compute_error);      // the one-pass iteration of kernels_onepass.hip
	Status onepass_check();                          // host sync: did a one-pass launch give up?
	Status materialize_w(bool whole_panel = true);
	Status ensure_w_rows();                          // fold the pending column scale into Wt_
	Status normalize_w(bool from_gram_partials, int norm_parts);   // column normalisation of W after its update
	Status normal_inverse(T* A, T offdiag, T diag);  // Qinv_ <- (A + regulariser)^-1, A destroyed
	Status normal_inverse_fork(T* A, T offdiag, T diag);   // the same on the side stream; normal_inverse_join() before Qinv_ is read
	Status normal_inverse_join();
	bool passengers_ride(const FactorProductPlan& plan) const; // split-operand product, rank 64: the Gram passengers find CUs beside the product blocks
	bool inverse_rides(const FactorProductPlan& plan) const;   // the inverse can be a passenger workgroup of the product launch
	Status finish_upload(T* Vcol);
	Status upload_triplets(std::vector<int>& rows, std::vector<int>& cols, std::vector<T>& vals);   // sparse mode: builds CSR + CSC on the host (the fall-back of upload_sparse_device)
	Status upload_sparse_device(int format, const T* values, const int* a, const int* b, long nnz, int base, bool* fallback);   // ... on the device (kernels_sparse_setup.hip)
	Status setup_kl_blocks();
	bool sparse_setup_on_device_ = false;
	Status iterate_kl(bool compute_error);            // KL-divergence multiplicative update (sparse mode)
	Status fetch_error_terms(int count_n);            // enqueue the copies, do not wait
	void finalize_error(bool resolve);
	void record_begin(int kind = 0);
	void record_end();
	bool timed_launch_events(int kind, hipEvent_t* start, hipEvent_t* stop);      // a sampled launch that takes its own start / stop events (hipExtLaunchKernel); then timed_launch_done()
	void timed_launch_done() { ev_used_ += 2; }

	int m_, n_, r_, RP_, alg_;
	int row_blocks_ = 1;
	AlgorithmParams prm_;
	long mpad_, npad_;
	long strideV_ = 0, strideVt_ = 0, elemsV_ = 0, elemsVt_ = 0;   // x-tiled images of V / Vt
	int num_cus_ = 256;
	hipStream_t stream_ = nullptr;
	// side stream for the r x r inverse of the least-squares algorithms: one workgroup, independent of the product
	// against V that follows the Gram matrix, so the two run side by side (the product leaves CUs idle)
	hipStream_t aux_ = nullptr;
	hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
	bool overlap_inverse_ = false;
	bool tiled_ = false;                      // V_/Vt_ are x-tiled (fp32 MFMA path)
	const char* last_error_ = "";

	T *V_ = nullptr, *Vt_ = nullptr;
	T *Wt_ = nullptr, *H_ = nullptr;          // factor panels
	T *Ws_ = nullptr, *Hs_ = nullptr;         // nsNMF: smoothed copies (W S)^T and S H
	T *slabs_ = nullptr;
	T *numW_ = nullptr;                       // reduced (V H^T)^T panel (LS family error terms, sharded runs)
	T *Wold_ = nullptr;                       // LS family: W before the update
	T *rowdot_part_ = nullptr;                // ... and the per-workgroup partial vectors of launch_row_dot's coalesced form
	T *G_ = nullptr, *G2_ = nullptr, *HHt_ = nullptr, *Qinv_ = nullptr, *gram_part_ = nullptr;
	T *sumsq_part_ = nullptr;
	// bf16-operand products (kernels_bf16.hip): fragment-ordered bf16 images of V, Vt and of the two factor panels
	bool bf16_ = false;
	void *Vb_ = nullptr, *Vtb_ = nullptr, *Wtb_ = nullptr, *Hb_ = nullptr;
	int ksW_ = 0, ksH_ = 0;
	FactorProductPlan planHb_, planWb_;
	// fp32 products on the bf16 matrix pipe by exact 3-way operand splitting (kernels_x3.hip): the fp32 tiled
	// images of V stay as they are, the factor panel is split into Wx3_ / Hx3_ before each product
	bool x3_ = false;
	// one resident image of V instead of two: W^T V reads the image tiled along its reduction index (the y-tiled form of
	// k_factor_product_x3; ~20 % slower per launch, half the memory).  Chosen when two images would not fit, or by
	// NMFAMD_ONE_IMAGE.
	bool one_image_ = false;
	int img_th_ = 128;        // rows per tile of the V image: 16 with one resident image (both kernel forms read contiguously), else the plan's
	bool wx3_valid_ = false, hx3_valid_ = false;   // Wx3_ / Hx3_ hold the split image of the current Wt_ / H_
	void *Wx3_ = nullptr, *Hx3_ = nullptr;
	void* qx3_ = nullptr;     // split image of the r x r operand of the wide fp32 panel update
	FactorProductPlan planHx_, planWx_;
	// sparse-V compute path (kernels_sparse.hip): CSR and CSC images of V, 0-based
	bool sparse_ = false;
	long nnz_ = 0;
	int *csr_ptr_ = nullptr, *csr_idx_ = nullptr, *csc_ptr_ = nullptr, *csc_idx_ = nullptr, *csc_from_csr_ = nullptr;
	T *csr_val_ = nullptr, *csc_val_ = nullptr, *q_ = nullptr, *q2_ = nullptr;
	T *t_vwh_ = nullptr, *t_kl_ = nullptr, *rowsum_part_ = nullptr, *sW_ = nullptr, *sH_ = nullptr;
	// KL half-steps with the gathered factor cut into blocks that fit an XCD's L2 (kernels_sparse.hip, k_kl_fused): blocked pointer arrays
	// [rows][blocks + 1] of the CSR image (W step: column blocks of H) and of the CSC image (H step: row blocks of W), partial panels
	int kl_blocks_w_ = 1, kl_blocks_h_ = 1;
	int *csr_bptr_ = nullptr, *csc_bptr_ = nullptr;
	T *kl_part_ = nullptr, *kl_tpart_ = nullptr;
	double sum_v_ = 0, kl_ = 0;
	// KL error terms travel like the Frobenius ones: pinned landing buffer [t_vwh (m) | t_kl (m) | sW (RP) | sH (RP) | psR (RP)],
	// copied stream-ordered, summed on the host only when somebody reads the error
	T* pin_kl_ = nullptr;
	bool kl_pending_ = false, kl_unresolved_ = false;
	std::vector<T> h_klrow_, h_sW_, h_sH_;
	// rank-64 MU fast path: W is kept unnormalised with a pending column scale (kernels_mu64.hip)
	float *gramW_part_ = nullptr, *gramH_part_ = nullptr, *scale_ = nullptr, *Graw64_ = nullptr;
	bool w_col_split_ = false;       // V H^T from 128 x 32 workgroups and one slab (narrow column shards, Engine::init)
	bool gram_spread_ = false;       // ... or, one slice: the sixteen passengers share the ten tiles' K ranges (gram_image.h, spread form) and the last one finishes G_
	unsigned* gram_spread_counter_ = nullptr;
	int gram_ksplit_ = 1;            // K slices of the W^T W passengers (gram_image.h): > 1 for column shards narrower than config 2
	float* Gpart_ = nullptr;         // [GRAM_KSPLIT_MAX][4096] unscaled slices of W^T W (the H update adds and scales them, and stores G_)
	float* wsq_part_ = nullptr;      // [mpad / 32][64] partial sums of squares of the rows the last W update wrote (k_mu64_update32<true>): the pending column scale's source
	bool fused_ready_ = false, w_pending_ = false;
	bool h_product_ahead_ = false;   // begin_next_iteration() has enqueued the W^T V launch of the next iteration_mu64 (cleared by whatever changes W, H or V)
	// split-operand path: Gram matrices from the split images (gram_image.h), 32-column update kernel -- no partial Gram matrices
	bool gram_image_ = false;
	GramReduceArgs gram_args(bool of_w, float* G, float* scale, int normalize) const;
	Status standalone_gram(const GramReduceArgs& rg);
	Status mu64_update(bool is_w, const T* slabs, int S, long slab_stride, const T* Q, bool compute_error, const PeerSlabs* peers = nullptr);
	// generic rank-64 fp32 path: the update kernel leaves partial Gram matrices of what it wrote (gram_from_update())
	bool gram_w_ready_ = false;      // G_ holds W^T W of the current (normalised) W
	// padded rank 256 with bf16 product operands (kernels_tri.hip): one pass per factor between its update and the product that streams
	// it (normalise + smooth + bf16 fragments), Gram matrices of the smoothed panels from the unsmoothed ones (S G S)
	bool tri_ = false;
	unsigned* tri_ride_counters_ = nullptr;          // arrival counters of the Gram passengers (tri_gram_tile.h): zero between launches
	bool tri_ride_w_ = false, tri_ride_h_ = false;   // the Gram matrix of W / of S H rides in the W^T V / V (S H)^T launch (Engine::init leaves TRI_PASSENGERS CUs free)
	bool wtb_valid_ = false;         // Wtb_ holds the bf16 fragments of the current W as it lies in Wt_ (unsmoothed; without the pending column scale)
	bool tri_scale_pending_ = false; // W = Wt_ diag(d), d(c) = 1 / sqrt(staged sums in colsq_): the column normalisation of the last W update has not been folded into the panel
	bool tri_scale_from_gram_ = false; // ... and its sums of squares are still to come out of the next Gram reduction (tri_prepare_w), into colsq_
	bool sole_rank_ = false;
	T* kl_scale_ = nullptr;          // KL iteration: W's pending column scale d (RP factors, from k_kl_sums) ...
	bool kl_scale_pending_ = false;  // ... Wt_ holds the unnormalised result of the last KL W update (materialize_w folds d in)
	bool kl_sw_ready_ = false;       // sW_ holds the column sums of the current W (left by its last KL update; a W set from outside needs the pass over the panel)
	bool kl_err_iter_ = false;       // sharded KL: h_step's compute_error, for the W-side evaluation in w_products
	long err_total_columns_ = 0;     // sharded runs: columns of the whole matrix (0: this engine's own n)
	Status kl_h_step();
	Status kl_w_products(T* exchange, bool compute_error);
	Status kl_w_finish(const T* exchange, bool compute_error);
	bool tri_w_den_bf16_ = true;     // the W update's r x r product takes the old rows rounded to bf16 (NMFAMD_TRI_FP32_DEN=1: six-term fp32-accurate product)
	bool hb_valid_ = false;          // Hb_ holds the bf16 fragments of the current smoothed H (written by the H update)
	// Gw_raw_ holds W^T W without the pending scale: what the error term's trace multiplies it with
	const T* tri_trace_scale() const { return (tri_ && tri_scale_pending_) ? reinterpret_cast<const T*>(colsq_) : nullptr; }
	bool tri_gw_ready_ = false;      // Gw_raw_ / G_ describe the current W
	bool tri_rows_cover_ = false;    // the last w_normalize_rows() covered every row of W
	bool w_rows_stale_ = false;      // row-block sharded run with the fragment exchange: the fp32 rows of the other ranks' blocks in Wt_ are those of an earlier iteration
	std::function<Status()> w_gather_hook_;
	int colsq_parts_ = 1;            // staged partial vectors in colsq_ (kernels_tri.hip: launch_colsq_stage)
	float *gram_tri_part_ = nullptr, *Gw_raw_ = nullptr, *Gh_raw_ = nullptr, *colsq_ = nullptr;
	bool qx3_holds_g_ = false, qx3_holds_hht_ = false;   // qx3_ holds the split image of G_ / of the smoothed H H^T (k_smooth_gram)
	void tri_smoothing(T* offdiag, T* diag) const;
	Status tri_prepare_w(GramReduceArgs* ride = nullptr);          // Wtb_, Gw_raw_, G_ for the H step (ride: the Gram matrix as passengers of the product launch that follows, tri_gram_tile.h)
	Status tri_prepare_h(T* hht, bool local_q, GramReduceArgs* ride = nullptr);    // Hb_, Gh_raw_, hht (= the Gram matrix of the smoothed H) for the W step
	Status tri_update_w(const T* num, int S, long stride, const T* hht);   // W update + normalisation + everything tri_prepare_w() would do
	// fused double-precision iteration (round 6, iterate_fused64): Wt_ holds the UNNORMALISED result of the last W update, sumsq_part_ its per-workgroup sums of squares;
	// the column scale is applied by the consumers (kernels_f64.hip: gram_ride_f64 -> f64_scale_, PanelFusedF64) -- materialize_w() folds it into the panel
	bool f32w_pending_ = false;                      // fused fp32 iteration at padded ranks >= 128: Wt_ unnormalised, sumsq_part_ holds its sums of squares, Wx3_ its split image
	float* f32w_scale_ = nullptr;                    // ... the pending column scale (k_gram_reduce_x3)
	int f32w_ride_h_ = 0, f32w_ride_w_ = 0;         // ... K slices per super-block of W^T W / H H^T that ride the product launches as passenger workgroups (0: a launch of their own)
	bool f64_pending_ = false;
	bool f64_product_ahead_ = false;                 // begin_next_iteration() has enqueued launch 1 of the next iterate_fused64 (cleared by whatever changes W, H or V)
	Status fused64_product_h();
	double *f64_scale_ = nullptr, *f64_partial_ = nullptr;
	unsigned* f64_counters_ = nullptr;
	int *f64_items_h_ = nullptr, *f64_items_w_ = nullptr;   // XCD-aware work tables of the passengers (gram_ride_f64_items)
	int f64_slices_h_ = 0, f64_slices_w_ = 0;        // K slices of the Gram passengers riding in the W^T V / V H^T launch
	const GramRideF64* ride64_ = nullptr;            // set around a product_h / product_w call
	unsigned long long* f64_stamps_ = nullptr;       // measurement builds (NMFAMD_F64_STAMPS = file): [4 launches][4096][8] stamps of the LAST fused iteration, written out by the destructor
	bool gram_h_partials_ = false;   // gramH_part_ describes the current H
	bool h_partials_unneeded_ = false; // set by iterate() around its H step: GDCLS takes H H^T from the split image beside the product against V
	int normalize_next_ = 0;
	T *psN_ = nullptr, *psR_ = nullptr;
	long ps_stride_ = 0;
	double* inv_work_ = nullptr;
	T* stage_ = nullptr;                      // upload/download staging (max(m, n) x r)
	int* range_flag_ = nullptr;               // device word, see launch_column_sumsq
	long slab_stride_ = 0;
	int gram_parts_ = 128;
	FactorProductPlan planH_, planW_;
	// one pass over V per iteration (kernels_onepass.hip): scratch of the in-launch hand-offs, per-XCD partial results
	bool one_pass_allowed_ = true, one_pass_ = false, one_pass_gave_up_ = false;
	void *op_part_ = nullptr, *op_hfrag_ = nullptr;
	unsigned *op_ctl_ = nullptr;                      // [0..7] tickets, [8] abort flag
	float *op_slabs_ = nullptr, *op_hh_part_ = nullptr, *op_ps4_ = nullptr;
	T* op_H2_ = nullptr;                              // the panel the next one-pass launch writes the new H into (swapped with H_ after it)
	unsigned op_seq_ = 0;
	unsigned* pin_abort_ = nullptr;
	unsigned long long* op_stamps_ = nullptr;          // diagnostic builds (NMFAMD_ONEPASS_STAMPS = file the last launch's stamps go to)

	T *pin_psN_ = nullptr, *pin_psR_ = nullptr;
	bool error_terms_stay_ = false;
	bool ps_last_direct_ = false;   // ... and the most recent error iteration did (error_terms_to_device reads the pinned buffer then)
	bool ps_direct_ = false;        // this (error) iteration's update kernels write the pinned buffer directly (iterate_mu64)
	T* pin_psN_dev_ = nullptr;      // the device's address of pin_psN_ (hipHostGetDevicePointer): fetch_error_terms writes it from a kernel
	hipEvent_t err_event_ = nullptr;
	bool err_pending_ = false, err_unresolved_ = false;
	int err_count_ = 0;
	std::vector<T> h_vtv_, h_psN_, h_psR_;
	double frob_ = 0, frob2_ = 0, rmsd_ = 0;

	bool timing_ = false;
	int timing_stride_ = 1;
	long timing_iter_ = 0;
	bool timing_now_ = false;
	std::vector<hipEvent_t> ev_;
	std::vector<char> ev_kind_;              // per event pair: which product it brackets (record_begin)
	size_t ev_used_ = 0;
};

// Host-side part of the error evaluation: sorted, interleaved accumulation in double
// (source/nmf/FrobeniusResolver.cpp:29-51).  The two iteration-dependent vectors are sorted here.
template <typename T>
double resolve_frobenius(const std::vector<T>& vtv_sorted, std::vector<T>& htwtv, std::vector<T>& hhtwtw);
// ... before the root (negative for a term vector that is not an error at all: the ALS family's constant-W trace)
template <typename T>
double resolve_frobenius_squared(const std::vector<T>& vtv_sorted, std::vector<T>& htwtv, std::vector<T>& hhtwtw);

} // namespace nmfamd
