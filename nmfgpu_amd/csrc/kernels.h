// kernels.h -- launcher prototypes of the gfx950 kernels in kernels.hip (internal header).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <vector>

namespace nmfamd {

// How one factor product OUT(c, x) = sum_y F(c, y) A(x, y) is cut into workgroups.
struct FactorProductPlan {
	int th;           // x-tile height: 128 or 160 rows
	int xtiles;       // tiles of the output index x
	int steps_total;  // reduction length in MFMA K-steps (two y per step)
	int splits;       // workgroup slices of the reduction range = number of output slabs
	int nb;           // 32-wide N-blocks per wave tile: 2, or 1 when only the first 32 panel columns are needed (rank <= 32)
	int chunks;       // launches needed to cover RP = chunks * 64 factor rows
	int xhalves = 0;    // fp64 product: 64-row halves of x-tiles that hold valid rows
	int half_tiles = 0; // fp64 product: 1 = a workgroup takes one 64-row half of an x-tile and eight pieces of its K slice (kernels_f64.hip, RH = 1): small grids
	int col_split = 0; // split-operand product, RP = 64 only: 2 = a workgroup takes 32 of the 64 panel columns (twice the workgroups, half the MFMAs per K-step and
	                   // wave; the operand is split twice) -- for small reduction ranges, where the 128 x 64 form leaves most of the chip idle; same bits
};

FactorProductPlan plan_factor_product(int X, int Y, int RP, int num_cus);

// Optional passenger of a factor-product launch (rank-64 MU fast path, kernels_mu64.hip): reduce
// `parts` partial 64 x 64 Gram matrices into G, optionally turning its diagonal into column scales.
// Raises a kernel's dynamic LDS limit once per DEVICE (the attribute is per device; a process may switch
// devices through nmfgpu::chooseGpu).  `done` is the caller's static bit mask, one bit per device ordinal.
// (several threads may compute at once -- per-thread contexts, rank threads of a "numGpus" team: the mask is atomic; a lost
// race only repeats the idempotent hipFuncSetAttribute)
inline hipError_t allow_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
	e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
	if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
	return e;
}

struct GramReduceArgs {
	const float* partials;  // [parts][4096]
	int parts;
	float* G;               // [4096]
	float* scale;           // [64] or nullptr
	int normalize;          // 1: scale = 1/sqrt(diag), G scaled on both sides; 0: scale = 1
	// the other kind of passenger (least-squares algorithms; fp32 kernel only, not together with the reduction):
	// inv_out <- (inv_a + regulariser)^-1, 64 x 64, one workgroup running k_inverse_gj64's body
	const float* inv_a = nullptr;
	float* inv_out = nullptr;
	float inv_offdiag = 0.f, inv_diag = 0.f;
	int inv_r = 0;
	// third kind (split-operand product only): G straight from the panel's split image instead of from partial matrices
	// (gram_image.h); image_ks = K-steps of 16 panel rows in the image (the all-zero step that closes it not counted)
	const void* image = nullptr;
	int image_ks = 0;
	// ... with normalize: the column scales from per-workgroup partial sums of squares the update kernel left ([colsq_parts][64], gram_image.h) instead of
	// from the diagonal of the product
	const float* colsq_part = nullptr;
	int colsq_parts = 0;
	// ... cut into ksplit slices of the K range (gram_image.h): 10 * ksplit passenger blocks, G then receives ksplit UNSCALED partial matrices ([ksplit][4096])
	// that the consumer adds and scales (launch_mu64_update32, qsplit); with normalize the scales must come from colsq_part
	int ksplit = 0;
	// ... or (spread > 0, the sixteen passengers of a whole problem): the ten tiles' K ranges dealt evenly to all GRAM_REDUCE_BLOCKS workgroups; G receives up to
	// GRAM_SPREAD_SLICES unscaled pieces per tile ([GRAM_SPREAD_SLICES][4096], absent pieces stay zero: the buffer is cleared once)
	int spread = 0;
	// ... and the LAST of the sixteen to finish adds the pieces in order and scales them as that consumer would -- (sum * scale[column]) * scale[row] -- into
	// spread_out (the finished 64 x 64 matrix); spread_counter: one word, zero between launches.  Its chain ends long before the product's, so the consumer gets one
	// finished matrix (the H update with three pieces to add took 6.8 us against 5.6)
	float* spread_out = nullptr;
	unsigned* spread_counter = nullptr;
	// fourth kind (bf16 factor product at padded rank 256 only, tri_gram_tile.h): the 256 x 256 Gram matrix of a panel from its bf16 fragments (tri_frags, tri_ks K-steps
	// of 16 panel rows) by TRI_PASSENGERS workgroups (K slices, the last one of a half reduces): G (fp32, both triangles), tri_diag (its diagonal, or nullptr), tri_x3 (its split image, or nullptr)
	const void* tri_frags = nullptr;
	int tri_ks = 0;
	void* tri_x3 = nullptr;
	float* tri_diag = nullptr;
	float* tri_partial = nullptr;       // [TRI_PASSENGERS / 2][36][1024] partial tiles
	unsigned* tri_counters = nullptr;   // two counters, zero between launches
	// fifth kind (split-operand product at padded ranks 128 ... 512, gram_wide.h): wide_parts K slices of every 128 x 128 super-block of the Gram matrix of the
	// fp32 panel wide_P (wide_len valid rows) as passenger workgroups -- what k_gram_wide_x3 does as a launch of its own; wide_partial: [wide_parts][RP][RP]
	// (blocks on and above the diagonal), reduced by launch_gram_reduce_x3 behind the product launch
	const float* wide_P = nullptr;
	int wide_len = 0, wide_parts = 0;
	float* wide_partial = nullptr;
};
constexpr int GRAM_REDUCE_BLOCKS = 16;
constexpr int TRI_PASSENGERS = 32;                 // tri_gram_tile.h: 16 K slices x 2 halves of the 36 upper-triangle tiles
constexpr int GRAM_IMAGE_TILES = 10;               // gram_image.h: upper triangle of the 4 x 4 grid of 16 x 16 tiles
constexpr int GRAM_SPREAD_SLICES = 3;              // ... pieces a tile arrives in when ten tiles are dealt to sixteen workgroups (gram_image.h, spread form)
constexpr int GRAM_KSPLIT_MAX = 8;                 // ... and the most K slices its K-split form is cut into

// X: valid output length (the plan picks the tile height; panels must be allocated for xtiles * th rows).
// A is x-TILED: A(x, y) at A[(x/th) * tile_stride + y*th + x%th] (launch_tile / launch_tile_transposed).
// slabs: plan.splits partial results, slab s at slabs + s * slab_stride, panel layout [x][RP].
hipError_t launch_factor_product_f32(const FactorProductPlan& p, const float* A, long tile_stride, const float* F, int RP,
                                     float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg = nullptr);

// Column-major (ld) -> x-tiled copy of an X x Y matrix into a ZERO-FILLED destination (untile: the reverse),
// and the x-tiled image of the TRANSPOSE of a column-major I x J matrix.
template <typename T>
hipError_t launch_tile(const T* src, long ld, int X, int Y, T* dst, long tile_stride, int th, bool untile, hipStream_t stream);
template <typename T>
hipError_t launch_tile_transposed(const T* src, long ld, int I, int J, T* dst, long tile_stride, int th, hipStream_t stream);

// Diagnostic build of the same kernel with in-kernel clock stamps (8 x uint64 per wave:
// shader clock at entry / first MFMA / loop end / kernel end, 100 MHz real time at entry / end, steps, XCC id).
hipError_t launch_factor_product_f32_stamped(const FactorProductPlan& p, const float* A, long lda, const float* F, int RP,
                                             float* slabs, long slab_stride, unsigned long long* stamps, hipStream_t stream);

// Rank-64 fp32 multiplicative-update fast path (kernels_mu64.hip).
// Partial Gram matrices of a panel, one per 64 panel columns ([len_pad/64][4096]), no reduction.
hipError_t launch_mu64_gram_partials(const float* P, int len_pad, float* partial, hipStream_t stream);
// Stand-alone form of the Gram reduction (used outside the iteration loop).
hipError_t launch_mu64_gram_reduce(const GramReduceArgs& rg, hipStream_t stream);
// Fused update (slab reduction, r x r product, element-wise update, error terms, partial Gram of the
// result).  is_w = 0: H panel, numerator scaled by `scale`; is_w = 1: Wt panel kept UNNORMALISED,
// old values scaled by `scale`; ps = tr(H^T W^T V) terms (H) / tr(H H^T W^T W) terms (W, needs Gprev).
hipError_t launch_mu64_update(int is_w, float* P, const float* slabs, int S, long slab_stride, const float* Q, const float* scale,
                              float eps, float* ps, int len_valid, int len_pad, float* gram_partial, const float* Gprev,
                              int compute_error, hipStream_t stream, void* x3_out = nullptr, int x3_ks = 0);
// The same update on 32-column tiles WITHOUT the partial Gram matrices (twice the workgroups, half the serial MFMA work per
// workgroup, no Gram in the critical path): for callers that take the Gram matrices from the split image (gram_image.h).
// x3_out is mandatory here.
// peers (optional): the summands are peers->count panels given by pointer (the exchange panels of a column-sharded run's ranks, in rank order; on this
// device or in a peer's memory) instead of S slabs slab_stride apart
constexpr int PEER_SLABS_MAX = 16;
struct PeerSlabs { const float* p[PEER_SLABS_MAX]; int count; };
// nsNMF's smoothing matrix S = (diag - off) I + off 1 1^T over the first r rank rows, applied AROUND the H update's r x r product (k_mu64_update32<.., NS>)
struct SmoothAround { float off, diag; int r; };
hipError_t launch_mu64_update32(int is_w, float* P, const float* slabs, int S, long slab_stride, const float* Q, const float* scale,
                                float eps, float* ps, int len_valid, int len_pad, const float* Gprev, int compute_error, hipStream_t stream,
                                void* x3_out, int x3_ks, const PeerSlabs* peers = nullptr, float* colsq_part = nullptr, int qsplit = 0, float* q_out = nullptr,
                                const SmoothAround* ns = nullptr);
// (ns, H update only: num = S D sum(slabs), den = S Q (S h), the split image written is that of S h_new -- the operand of nsNMF's V (S H)^T; the panel keeps h_new)
// (qsplit > 1, H update: Q holds qsplit unscaled partial matrices 4096 elements apart -- the K-split Gram passengers' output, GramReduceArgs::ksplit; every
//  workgroup adds them in order and scales the sum by `scale` on both sides, workgroup 0 also stores the finished matrix to q_out)
// (colsq_part, W update only: len_pad / 32 vectors of 64 partial sums of squares of the new rows, one per workgroup -- what the next Gram passengers turn into
//  the pending column scale, GramReduceArgs::colsq_part)
// out[i] = sum_k src.p[k][i] (rank order), count a multiple of 4: the r x r parts of the ranks' exchange buffers
hipError_t launch_sum_peers(const PeerSlabs& src, float* out, int count, hipStream_t stream);
// G (64 x 64) = P P^T from the split image of P (image_ks K-steps); normalize / scale as in GramReduceArgs.  Stand-alone form
// of the passenger workgroups of the split-operand product launch.
hipError_t launch_gram_from_image(const void* image, int image_ks, float* G, float* scale, int normalize, hipStream_t stream, const float* colsq_part = nullptr, int colsq_parts = 0);
hipError_t launch_gram_image_args(const GramReduceArgs& rg, hipStream_t stream);      // ... every field of rg as given, the K-split form included
// P(c, y) *= scale(c)
hipError_t launch_mu64_apply_scale(float* P, int len_pad, const float* scale, hipStream_t stream, void* x3_out = nullptr, int x3_ks = 0);
// G <- sum of `parts` partial 64 x 64 matrices; with scale != nullptr also scale(c) = 1 / sqrt(G(c, c)) (1 if 0) and
// G <- diag(scale) G diag(scale): the stand-alone (not passenger) form of the Gram reduction
hipError_t launch_gram64_from_partials(const float* partials, int parts, float* G, float* scale, hipStream_t stream);
hipError_t launch_gram64_normalize_all(const float* partials, int parts, float* Graw, float* G, float* scale, float* P, int len_pad, void* x3_out, int x3_ks, hipStream_t stream);
// the same in ONE launch: column scales from the update kernel's sums of squares (sq_parts vectors of 64) instead of the reduced diagonal; G = D (sum of the partial
// matrices) D; P <- P D and its split image
hipError_t launch_gram64_reduce_scale_all(const float* gram_part, int parts, const float* sumsq_part, int sq_parts, float* G, float* scale, float* P, int len_pad,
                                          void* x3_out, int x3_ks, hipStream_t stream);

// Generic (VALU) form, writes the finished panel (no slabs).  Xpad multiple of 64, RP multiple of 32.
template <typename T>
hipError_t launch_factor_product_valu(const T* A, long lda, int Xpad, int Y, const T* F, int RP, T* out, hipStream_t stream);

template <typename T>
hipError_t launch_reduce_slabs(const T* slabs, int S, long slab_stride, T* out, long count, hipStream_t stream);

// G = P P^T over len panel columns; partial holds parts * RP * RP elements of scratch.
template <typename T>
hipError_t launch_gram(const T* P, int RP, int len, int parts, T* partial, T* G, hipStream_t stream);

enum PanelMode { PANEL_MU = 0, PANEL_LS = 1, PANEL_SET = 2 };
// Extras of the multiplicative update at padded rank 256 with bf16 product operands (config 4, kernels_tri.hip).  W is carried unnormalised with a
// pending column scale D (as the rank-64 path does), and its bf16 fragments hold the unnormalised, unsmoothed values: scale and nsNMF smoothing move to
// the OUTPUT side of the product, (W D S)^T V = S D (W^T V) -- O(r n) work inside the H update instead of a pass over the panel.
struct PanelTriExtras {
	// A pending column scale is handed over as the staged sums of squares the update left behind (launch_colsq_stage: `parts` vectors of RP, added in
	// order): d(c) = sum > 0 ? 1 / sqrt(sum) : 1 -- kernel::normalizeColumns (KernelNormalizeColumns.cu:37-58) as a factor.  Every consumer forms d the
	// same way (tri_pending_scale() below), so they all see the same bits.  nullptr: ones.
	// numerator transform, per panel row y:  num'(y, c) = a * d(c) num(y, c) + b * sum_{c'} d(c') num(y, c'),  c < r  (zero beyond); d from num_colsq.
	bool num_transform = false;
	const float* num_colsq = nullptr;
	int num_colsq_parts = 0;
	float num_a = 1.0f, num_b = 0.0f;
	int r = 0;
	// denominator with the smoothing (and a pending scale) applied around the r x r product instead of inside its operand (32-row kernel):
	//   den(y, :) = S D G D S old(y, :),  S = den_a I + den_b 1 1^T on the first r columns, D = diag(d) from den_colsq (nullptr: ones), G = the split image given
	// -- the Gram matrix arrives as the reduction left it (unnormalised, unsmoothed) and no O(r^2) pass over it is launched
	bool den_transform = false;
	float den_a = 1.0f, den_b = 0.0f;
	const float* den_colsq = nullptr;
	int den_colsq_parts = 0;
	// the panel's own pending column scale: every old value is read as old(y, c) * d(c), d from old_colsq
	const float* old_colsq = nullptr;
	int old_colsq_parts = 0;
	// the old rows enter the r x r product rounded to bf16 (three MFMAs per tile instead of six, no operand split; Q keeps its three planes); the
	// element-wise step keeps the fp32 values.  Measured at rank 256 / nsNMF (tools/tri_accuracy.py): the bf16 mode's distance from the fp64 oracle does
	// not move (W 3.4e-4 vs 3.2e-4 after 10 iterations, 1.17e-3 vs 1.16e-3 after 40): it is set by the rounding of V and the factors in the big products
	bool old_as_bf16 = false;
	// bf16 fragments of the NEW rows (layout of k_finish_panel_bf16), K-steps >= frag_KS are not written; the 32-row kernel can smooth them on the way
	// (f(y, c) = frag_a x(y, c) + frag_b sum_c' x(y, c'), c < r; needs r), the 128-row kernel writes them as they are (frag_a = 1, frag_b = 0)
	void* frag_out = nullptr;
	long frag_KS = 0;
	float frag_a = 1.0f, frag_b = 0.0f;
};
int panel_update_rows(int RP, size_t elem);
// number of per-workgroup sum-of-squares partials launch_panel_update writes for a panel of len_pad columns
int panel_update_parts(int RP, size_t elem, int len_pad);

// fp32 / padded rank 64 specialisations (kernels_fast.hip); the generic launchers dispatch to them
hipError_t launch_gram64_f32(const float* P, int len, int parts, float* partial, float* G, hipStream_t stream);
hipError_t launch_panel_update64_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int len_pad,
                                     float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream);
// fp32 / padded rank 64, LDS-staged (kernels_wide.hip): 64 panel rows per workgroup
// gram_partial (optional): len_pad / 64 partial 64 x 64 Gram matrices of the new rows (layout of k_mu64_update)
hipError_t launch_panel_update64_lds_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int len_pad,
                                         float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream,
                                         float* gram_partial = nullptr, void* x3_out = nullptr, int x3_ks = 0);
// true when launch_panel_update<float> at padded rank 64 can deliver those partial Gram matrices
bool panel_update_delivers_gram(int RP, size_t elem);
// fp32 / padded rank 128 ... 512 (kernels_wide.hip)
bool panel_update_wide_available(int RP);
// long panels at padded rank 128 / 256 (kernels_wide.hip, k_panel_update_wide64_mu): multiplicative update with the r x r operand given as
// its split image; the result may go to another panel (P_out != P_in), e.g. to keep the unnormalised W readable while it is normalised
bool panel_update_long_available(int RP, int len_pad);
hipError_t launch_panel_update_long_mu(const float* P_in, float* P_out, const float* slabs, int S, long slab_stride, const void* q_split, int RP, int len_pad,
                                       float eps, float* ps, int len_valid, float* sumsq_part, hipStream_t stream, const PanelTriExtras* tri = nullptr);
bool gram_wide_available(int RP);
// len: valid panel rows (the padding rows behind them are zero); partial: parts * RP * RP elements
hipError_t launch_gram_wide_f32(const float* P, int RP, int len, int parts, float* partial, float* G, hipStream_t stream);
// q_split (optional): 3 * 16 * (RP / 16 + 1) * (RP / 32) * 64 bytes of scratch; when given, the r x r product runs on the bf16
// matrix pipe with exactly split operands (fp32 accuracy, kernels_x3.hip) instead of the fp32 MFMA instructions
// Extras of the fused fp32 iteration at padded ranks 128 ... 512 (Engine::iterate_fused32w; the float counterpart of PanelFusedF64)
struct PanelFusedF32 {
	const float* old_scale = nullptr;   // W update: every old value is read as old(y, c) * old_scale[c]
	int h_side = 0;                     // H update with W = Wt D (S): Q is the RAW Gram matrix of the panel Wt, and
	const float* scale = nullptr;       //   d (nullptr: ones):  num <- S D num,  den = S D Q D S old
	int smooth = 0;                     //   0: S = I
	float off = 0.f, diag = 1.f;
	int r = 0;
	float* smooth_out = nullptr;        // H update with smoothing: a second panel that receives S new(y, :)
	void* x3_out = nullptr;             // the split image (k_pack_panel_x3's layout) of what the next product multiplies with: S new where there is smoothing, else new
	long x3_ks = 0;                     // ... K-steps >= x3_ks are not written
};
hipError_t launch_panel_update_wide_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int RP, int len_pad,
                                        float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream,
                                        void* q_split = nullptr, const PanelTriExtras* tri = nullptr, const PanelFusedF32* fused = nullptr);
// G = P P^T, its split image (the update kernel's operand: launch_panel_update_wide_f32 with Q == nullptr) and, W side, the pending column scale -- two launches
// the second half of launch_gram_wide_fused_f32 on its own: the slices came from passenger workgroups of a product launch (GramReduceArgs::wide_P)
hipError_t launch_gram_reduce_x3(const float* partial, int parts, int RP, float* G, void* qx3, const float* sumsq_part, int sq_parts, float* scale_out, hipStream_t stream);
// the slice count launch_gram_wide_fused_f32 settles on (at most `parts`)
int gram_wide_fused_parts(int RP, int len, int parts);
hipError_t launch_gram_wide_fused_f32(const float* P, int RP, int len, int parts, float* partial, float* G, void* qx3, const float* sumsq_part, int sq_parts,
                                      float* scale_out, hipStream_t stream);
template <typename T>
hipError_t launch_reduce_partials(const T* partial, int parts, long stride, T* out, long count, hipStream_t stream);
template <typename T>
hipError_t launch_normalize_panel_v2(T* P, int RP, int len_pad, T* sumsq_part, int parts, hipStream_t stream);

// See k_panel_update.  sumsq_part needs (len_pad / panel_update_rows) * RP elements.
template <typename T>
hipError_t launch_panel_update(int mode, T* P, const T* slabs, int S, long slab_stride, const T* Q, int RP, int len_pad,
                               T eps, T* ps, int len_valid, T* sumsq_part, T* num_out, hipStream_t stream, T* gram_partial = nullptr,
                               void* x3_out = nullptr, int x3_ks = 0,    // x3_out: split image of the new panel (kernels_x3.hip); only where panel_update_delivers_gram()
                               void* q_split = nullptr,                  // scratch for the split image of Q (wide fp32 panels, see launch_panel_update_wide_f32)
                               const PanelTriExtras* tri = nullptr);     // fp32, padded rank 256 only

// sumsq_part: parts * RP partial sums followed by 16 * RP elements of scratch
template <typename T>
hipError_t launch_normalize_panel(T* P, int RP, int len_pad, T* sumsq_part, int parts, hipStream_t stream);

template <typename T>
hipError_t launch_smooth_panel(const T* P, T* out, int RP, int r, long len_pad, T offdiag, T diag, hipStream_t stream);

template <typename T>
hipError_t launch_trace_small(const T* A, const T* B, int RP, int r, T* ps, hipStream_t stream, const T* b_colsq = nullptr, const T* b_scale = nullptr);

template <typename T>
hipError_t launch_row_dot(const T* A, const T* B, int RP, int r, long len, T* ps, hipStream_t stream, T* part = nullptr);      // part: ROW_DOT_GROUPS * RP scratch -> coalesced form
constexpr int ROW_DOT_GROUPS = 128;

template <typename T>
hipError_t launch_fill_small(T* A, int RP, int r, int reuse, T offdiag, T diag, hipStream_t stream);
// dst (device-visible host memory or device memory) <- src, count elements, by a small kernel on `stream`
template <typename T>
hipError_t launch_copy_small(T* dst, const T* src, long count, hipStream_t stream);
template <typename T>
hipError_t launch_copy_two(T* dst, const T* a, long na, const T* b, long nb, hipStream_t stream);      // dst = [a | b]

// Ainv = (A + regulariser)^-1, regulariser = offdiag everywhere, diag on the diagonal.  work: 2 * r * r doubles
// (r > 64 only; that route also adds the regulariser to A in place).
template <typename T>
hipError_t launch_inverse_small(T* A, int RP, int r, T* Ainv, double* work, T offdiag, T diag, hipStream_t stream);

template <typename T>
hipError_t launch_transpose(const T* src, long lds, int rows, int cols, T* dst, long ldd, hipStream_t stream);

// range_flag (optional, fp32): bit 0 is set when V holds a value outside what the split-operand product (kernels_x3.hip) is
// exact for -- not finite, |v| > 2^126 (the first bf16 cut would round to infinity) or 0 < |v| < 2^-100 (the third term would
// fall into the flushed bf16 subnormals)
template <typename T>
hipError_t launch_column_sumsq(const T* V, long ldv, int rows, int cols, T* ps, hipStream_t stream, int* range_flag = nullptr);

// format: 1 CSR (ptr = rowPtr, idx = column indices), 2 CSC (ptr = columnPtr, idx = row indices),
// 3 COO (idx = row indices, idx2 = column indices); outer = rows (CSR) / columns (CSC).
template <typename T>
hipError_t launch_densify(int format, const T* values, const int* ptr, const int* idx, const int* idx2, long nnz, int outer, int base,
                          T* V, long ldv, int rows, int cols, hipStream_t stream);

// element (c, y) = draw number (y + y_first) * r + c of the counter-based stream (y_first: a column shard's first global column)
template <typename T>
hipError_t launch_fill_uniform(T* P, int RP, int r, long len, long len_pad, uint64_t seed, hipStream_t stream, long y_first = 0);

// ---- fp64 MFMA factor product (kernels_f64.hip): same x-tiled image of A (tile height 128), K-steps of four y ----
FactorProductPlan plan_factor_product_f64(int X, int Y, int RP, int num_cus);
// Passengers of the fp64 product launch (kernels_f64.hip, gram_ride_f64): the Gram matrix of a factor panel as it lies, K-sliced, reduced by the last arriver of every
// 64 x 64 super-block; and (W side) the pending column scale from the W update's sums of squares
struct GramRideF64 {
	const double* P;            // the panel (RP columns); nullptr: no passengers
	int len;                    // its valid rows (rows behind them up to the padded length are zero)
	int slices;                 // K slices per super-block
	double* partial;            // [slices][super-blocks][4096] scratch
	unsigned* counters;         // [super-blocks], zero between launches
	double* G;                  // the reduced matrix, both triangles
	const double* sumsq_part;   // the pending column scale's source: sumsq_parts vectors of RP partial sums of squares (nullptr: no scale passengers)
	int sumsq_parts;
	double* scale_out;          // [RP] d(c) = sum > 0 ? 1 / sqrt(sum) : 1
	const int* items;           // [super-blocks * slices] which (super-block, slice) passenger pid takes (gram_ride_f64_items; nullptr: pid order)
	int stop;                   // measurement builds (NMFAMD_RIDE64_STOP): 1 = passengers return at once, 2 = after their partial block, 3 = after counting in (results void)
	unsigned long long* stamps; // measurement builds: [4096][8] wall-clock stamps (product workgroups from 0, passengers from 2048), tools/stamp_f64.py
};
int gram_ride_f64_workgroups(int RP, int slices);
hipError_t launch_factor_product_f64(const FactorProductPlan& p, const double* A, long tile_stride, const double* F, int RP,
                                     double* slabs, long slab_stride, hipStream_t stream, const GramRideF64* ride = nullptr);
void gram_ride_f64_items(const FactorProductPlan& p, int RP, int slices, std::vector<int>& items);

// Extras of the fused double-precision iteration (Engine::iterate_fused64): W stays unnormalised in its panel with a pending column scale d (a vector the scale
// passengers of the W^T V launch leave), and nsNMF's smoothing S = (diag - off) I + off 1 1^T (first r entries) is applied where a row of the panel sits in LDS anyway
struct PanelFusedF64 {
	const double* old_scale;    // W update: every old value is read as old(y, c) * old_scale[c]
	int h_side;                 // H update with W = Wt D (S): Q is the RAW Gram matrix of the panel Wt, and
	const double* scale;        //   d (nullptr: ones):  num <- S D num  ((Wt D S)^T V = S D (Wt^T V)),  den = S D Q D S old  (D and S applied around the MFMA product)
	int smooth;                 //   0: S = I
	double off, diag;
	int r;
	double* smooth_out;         // H update: a second panel that receives S new(y, :) (the operand of V (S H)^T and of its Gram matrix)
	// W update on error iterations: ceil(trace_r / 4) extra workgroups behind the panel's compute the r terms of tr(H H^T W^T W) (k_trace_small's arithmetic:
	// trace_out[d] = sum_i A(i, d) B(d, i) f(d) f(i), f = trace_scale or ones) -- the launch of its own and the boundary around it are gone
	const double* trace_a;
	const double* trace_b;
	const double* trace_scale;
	double* trace_out;
	int trace_r;
	unsigned long long* stamps; // measurement builds: [workgroups][8] wall-clock stamps (wide kernel)
};
inline int panel_fused_f64_extra_workgroups(const PanelFusedF64* fx) { return (fx != nullptr && fx->trace_out != nullptr) ? (fx->trace_r + 3) / 4 : 0; }
// fp64 / padded rank 64 panel update on the fp64 MFMA pipe (32 panel rows per workgroup: len_pad / 32 norm partials)
hipError_t launch_panel_update64_f64(int mode, double* P, const double* slabs, int S, long slab_stride, const double* Q, int len_pad,
                                     double eps, double* ps, int len_valid, double* sumsq_part, double* num_out, hipStream_t stream, const PanelFusedF64* fused = nullptr);

// fp64 Gram matrix on the MFMA pipe (padded rank 64 or k * 128)
hipError_t launch_gram_f64(const double* P, int RP, int len, int parts, double* partial, double* G, hipStream_t stream);
// fp64 / padded rank 128 ... 512: 16 panel rows per workgroup
bool panel_update_wide_f64_available(int RP);
hipError_t launch_panel_update_wide_f64(int mode, double* P, const double* slabs, int S, long slab_stride, const double* Q, int RP, int len_pad,
                                        double eps, double* ps, int len_valid, double* sumsq_part, double* num_out, hipStream_t stream, const PanelFusedF64* fused = nullptr);

// ---- bf16-operand factor product (kernels_bf16.hip) ------------------------------------------
// Fragment-ordered bf16 images: streamed matrix (x-tiled by 128, KS = ceil(Y / 16) K-steps) and the factor
// panel [y][RP] (RP = 64 or a multiple of 128; 16 * KS * RP / 32 * 64 bytes).
hipError_t launch_pack_stream_bf16(const float* src, long ld, int X, int Y, bool transposed, void* dst, int xtiles, int KS, hipStream_t stream);
hipError_t launch_pack_panel_bf16(const float* P, int RP, int len, void* dst, int KS, hipStream_t stream);

// kernels_tri.hip -- factor-side passes at padded rank 256 with bf16 product operands (config 4)
bool tri_kernels_available(int RP);
// rows [row0, row0 + rows) of the panel P (multiples of 32): optional column normalisation in place (colsq = colsq_parts vectors of RP partial sums of squares
// over ALL rows, added in order; or nullptr), nsNMF smoothing with (offdiag, diag) in registers, bf16 fragments of the smoothed rows into the pack `dst`
// (K-steps >= KS are not written)
hipError_t launch_finish_panel_bf16(float* P, int RP, int r, long row0, long rows, const float* colsq, int colsq_parts, float offdiag, float diag, void* dst, long KS, hipStream_t stream);
// the update kernel's per-workgroup sums of squares (`parts` vectors of RP) -> colsq_stage_parts() staged vectors (and, final_sum != nullptr, their sum)
int colsq_stage_parts();
hipError_t launch_colsq_stage(const float* part, int RP, int parts, float* staged, float* final_sum, hipStream_t stream);
// G (RP x RP, both triangles, exactly symmetric) = P^T P over `len` panel rows; partial: gram_tri_partial_elems(max_parts) floats
hipError_t launch_gram_tri(const float* P, int RP, int len, int max_parts, float* partial, float* G, int num_cus, hipStream_t stream);
long gram_tri_partial_elems(int max_parts);
// the same matrix from the panel's bf16 fragments (KS K-steps of 16 rows, layout of k_finish_panel_bf16 without smoothing): ONE bf16 MFMA per tile and
// K-step, i.e. G = P~^T P~ of the ROUNDED panel P~ -- exactly the operand the product against V multiplies with, so numerator and denominator of the
// H update refer to the same matrix.  colsq != nullptr: G(r, c) *= d(r) d(c), the pending column scale of the panel (PanelTriExtras).
hipError_t launch_gram_tri_bf16(const void* frags, int RP, long KS, int max_parts, float* partial, float* G, const float* colsq, int colsq_parts, int num_cus, hipStream_t stream);
// the same product, reduced WITHOUT any scaling into G and, in the same launch, into the split image x3_out that the update kernels take as their r x r operand
// (launch_panel_update with Q = nullptr; A(c, k) = G(k, c)); diag_out[c] = G(c, c) = the sum of squares of column c of the rounded panel -- the pending
// column scale of that panel in the form PanelTriExtras takes it (one "staged" vector)
hipError_t launch_gram_tri_bf16_image(const void* frags, int RP, long KS, int max_parts, float* partial, float* G, void* x3_out, float* diag_out, int num_cus, hipStream_t stream);
// P(y, c) *= d(c) over `rows` panel rows (folds a pending column scale, given as staged sums of squares, into the panel); scale_out (optional): the RP factors
hipError_t launch_scale_panel_tri(float* P, int RP, long rows, const float* colsq, int colsq_parts, float* scale_out, hipStream_t stream);
// Gs = S G S, S = (diag - offdiag) I + offdiag 1 1^T on the first r rows / columns
// x3_out (optional): also the split image of Gs that the wide update kernels take as their r x r operand (launch_panel_update with Q = nullptr);
// its closing all-zero K-step is the caller's (written once)
hipError_t launch_smooth_gram(const float* G, float* Gs, int RP, int r, float offdiag, float diag, void* x3_out, hipStream_t stream);
// the finishing pass over ALL rows of a panel and its Gram matrix in one launch (+ the reduction); colsq != nullptr: P_out (!= P_in) <- the
// column-normalised panel, G = the Gram matrix of the normalised panel
hipError_t launch_finish_and_gram(const float* P_in, float* P_out, int RP, int r, long rows, int len, const float* colsq, int colsq_parts, float offdiag, float diag,
                                  void* dst, long KS, int max_parts, float* partial, float* G, int num_cus, hipStream_t stream);
hipError_t launch_factor_product_bf16(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                      float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg = nullptr);
// K-split of the bf16 product for `xtiles` x-tiles and KS K-steps at padded rank RP
int plan_splits_bf16(int xtiles, int KS, int RP, int num_cus);
// workgroups the bf16 product launches for a plan at padded rank 256 (what is left of the chip can carry TRI_PASSENGERS)
int bf16_product_workgroups(const FactorProductPlan& p);
#ifdef NMFAMD_DIAG_BUILD
void set_factor_product_bf16_stamps(unsigned long long* stamps);      // measurement build: the calling thread's next rank-256 product launch stamps its waves' lives (tools/stamp_bf16.py)
#endif

// ---- fp32 product by exact 3 x bf16 operand splitting (kernels_x3.hip), padded rank 64 ------------
// The streamed matrix is the x-tiled fp32 image (tile height 128); the factor panel is split into
// three bf16 planes in fragment order, KS K-steps plus one all-zero step: 3 * 16 bytes * (KS + 1) * (RP / 32) * 64.
hipError_t launch_pack_panel_x3(const float* P, int RP, int len, void* dst, int KS, hipStream_t stream);
hipError_t launch_factor_product_x3(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP,
                                    float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg = nullptr,
                                    unsigned long long* stamps = nullptr, bool y_tiled = false, int image_tile = 128,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// (ev_start / ev_stop: the launch's own start and stop times go to these events -- hipExtLaunchKernel -- instead of the caller recording two events around it:
//  a bracketed launch costs the stream ~12 us of barrier packets, rocprofv3 trace of bench.py's sampled iterations)
int plan_splits_x3(int xtiles, int KS, int num_cus, int reserve = 0);

// ---- one pass over V per multiplicative-update iteration at padded rank 64 (kernels_onepass.hip) -----------------------
// The H update is column-separable once W^T W is known (reference: RN2 = RR H, RN = W^T V, multiplyDivide per element,
// source/nmf/AlgorithmMultiplicativeFrobenius.h:176-191; KernelMultiplyDivide.cu:29-43): a 32-column panel of V is loaded
// ONCE, W^T V of the panel is reduced over its rows, the panel's columns of H are updated, and (V H^T) is accumulated
// (:240-241) while the panel is still on chip.  One persistent launch of 256 workgroups (see the kernel).
struct OnePassArgs {
	const float* V; long tile_stride;     // the 16-row tiled image: V(i, j) at V[(i / 16) * tile_stride + j * 16 + i % 16]
	const void* Wx3;                      // split image of the (unnormalised) W panel, NBT = 2 (kernels_x3.hip)
	const float* G;                       // 64 x 64: diag(scale) Wu^T Wu diag(scale)
	const float* scale;                   // 64: pending column scale of W
	const float* H;                       // panel [npad][64]: the old H
	float* H_out;                         // the new H (another panel: the four owner waves of a column read the whole old column at their own pace)
	float* ps; long ps_stride;            // error iterations: four partial vectors (one per owner wave) of the per-column terms of tr(H^T W^T V)
	float* slabs; long slab_stride;       // 8 partial (V H^T)^T panels [mpad][64], one per XCD
	float* hh_part;                       // [256][4096] partial H H^T, one per workgroup
	void* part_scratch;                   // [8][ONEPASS_SLOTS][32][256] x 64 B tagged partial sums
	void* hfrag_scratch;                  // [8][ONEPASS_SLOTS][64][4][8] x 8 B tagged split values of the new H columns
	unsigned* ticket;                     // [8] per-XCD arrival counters (never reset: 32 per launch)
	unsigned* abort_flag;                 // set when a workgroup gave up waiting (or found its XCD over-subscribed)
	int tile_rows;                        // mpad / 16
	int w_ks;                             // K-steps of 16 rows in the split image of W; step w_ks is the all-zero step that closes it
	int n;                                // valid columns
	int panels;                           // ceil(n / 32)
	unsigned seq;                         // launch number of this engine (tickets and tags derive from it)
	int compute_error;
	float eps;
	unsigned long long* stamps = nullptr; // diagnostic builds: 16 words per wave (see the kernel's end)
};
constexpr int ONEPASS_SLOTS = 4;
constexpr int ONEPASS_GROUP = 32;         // workgroups per XCD
constexpr int ONEPASS_XCDS = 8;
constexpr int ONEPASS_TILES_PER_WAVE = 5; // 16-row tiles per wave: 320 rows per workgroup, 10 240 per XCD
constexpr size_t onepass_part_bytes() { return (size_t)ONEPASS_XCDS * ONEPASS_SLOTS * ONEPASS_GROUP * 256 * 64; }
constexpr size_t onepass_hfrag_bytes() { return (size_t)ONEPASS_XCDS * ONEPASS_SLOTS * 64 * 32 * 8; }
// true when the shape fits the kernel's fixed cut (rows per XCD group) on this device
bool onepass_available(long mpad, int num_cus);
hipError_t launch_mu64_onepass(const OnePassArgs& a, hipStream_t stream);

// ---- sparse-V compute path (kernels_sparse.hip) ----------------------------------------------
// out(row, :) = sum_p val[p] P(idx[p], :) over the stored entries of `row`; rows in [rows, rows_pad) are zeroed.
template <typename T>
hipError_t launch_spmm_rows(const int* ptr, const int* idx, const T* val, const T* P, int RP, T* out, int rows, int rows_pad, hipStream_t stream);
// q[p] = val[p] / (A(row,:).B(idx[p],:) + eps) and the per-row sums of val*wh and val*log(val/wh)
template <typename T>
hipError_t launch_sddmm_quotient(const int* ptr, const int* idx, const T* val, const T* A, const T* B, int RP, T eps,
                                 T* q, T* t_vwh, T* t_kl, int rows, hipStream_t stream);
// KL half-step in one pass: out(row, :) = sum_p val[p] / (A(row,:).B(idx[p],:) + eps) * B(idx[p], :) (quotient and numerator
// together, one gather per stored entry); rows in [rows, rows_pad) are zeroed; optional per-row error terms as above
template <typename T>
// blocks > 1: ptr is the blocked pointer array [rows][blocks + 1] (a row's entries cut at the block boundaries of the gathered index);
// out then receives `blocks` partial panels out_stride apart (and t_vwh / t_kl `blocks` partial vectors of rows_pad), grid block-major
hipError_t launch_kl_fused(const int* ptr, const int* idx, const T* val, const T* A, const T* B, int RP, T eps,
                           T* out, T* t_vwh, T* t_kl, int rows, int rows_pad, hipStream_t stream, int blocks = 1, long out_stride = 0,
                           const T* a_scale = nullptr);      // a_scale (RP factors, optional): A(row, :) is read as A(row, c) * a_scale[c] -- W's pending column scale
template <typename T>
hipError_t launch_permute(const T* src, const int* perm, T* dst, long count, hipStream_t stream);
// sums(c) = sum_y P(c, y); partial: (len_pad / 128) * RP elements of scratch
template <typename T>
hipError_t launch_panel_rowsum(const T* P, int RP, int len_pad, T* partial, T* sums, hipStream_t stream);
// P(c, y) <- P(c, y) num(c, y) / (den(c) + eps); sumsq_part (optional): (len_pad / 128) * RP partial sums of squares
template <typename T>
// num: `parts` partial numerator panels part_stride apart, added in order
hipError_t launch_kl_update(T* P, const T* num, const T* den, int RP, int len_pad, T eps, T* sumsq_part, hipStream_t stream, int parts = 1, long part_stride = 0,
                            T* sum_part = nullptr,       // sum_part (optional): (len_pad / 128) * RP partial sums of the new values
                            const T* scale = nullptr);   // scale (RP factors, optional): P(c, y) <- (P(c, y) scale[c]) num(c, y) / (den(c) + eps)
// sums(c) = sum of the len_pad / 128 vectors of sum_part, divided by sqrt(sum of sumsq_part) where sumsq_part is given and that is > 0 (the column sums of a panel
// AFTER its column normalisation, from what its update left behind); RP in {64, 128, 256}
template <typename T>
hipError_t launch_kl_sums(const T* sum_part, const T* sumsq_part, int parts, int RP, T* sums, hipStream_t stream, T* scale_out = nullptr);      // scale_out (optional): 1 / sqrt(sum of squares), 1 where that is 0

// ---- the CSR and CSC images of a sparse V built on the device (kernels_sparse_setup.hip) ----------------------------------------
// flags (one int, zeroed by the caller): bit 0 = an entry outside the matrix or outside every pointer range, bit 1 = pointer array not ascending, bit 2 = the
// entries are not in (row, column) order.  format: 1 CSR (a = rowPtr, outer = rows), 2 CSC (a = columnPtr, outer = columns), 3 COO (a = rows, b = columns)
hipError_t launch_sp_expand(int format, const int* a, const int* b, long nnz, int outer, int base, int m, int n, int* row, int* col, int* flags, hipStream_t stream);
// ptr[0 .. segments] = exclusive scan of the histogram of key[0 .. count) (counts: `segments` ints of scratch), maxlen[0] = the longest segment
hipError_t launch_sp_histogram_scan(const int* key, long count, int segments, int* counts, int* ptr, int* maxlen, hipStream_t stream);
long sp_segment_sort_capacity();
// items <- the positions 0 .. count - 1 grouped by key (segment s at [ptr[s], ptr[s + 1])), every segment ascending by (minor[item], item) -- by item alone
// when minor == nullptr.  fill: `segments` ints of scratch; maxlen: the longest segment (<= sp_segment_sort_capacity())
hipError_t launch_sp_scatter_sort(const int* key, long count, int segments, const int* ptr, int* fill, int* items, const int* minor, int maxlen, hipStream_t stream);
template <typename T>
hipError_t launch_sp_gather_csr(const int* order, const int* row, const int* col, const T* val, long nnz, int* csr_idx, T* csr_val, int* rowq, hipStream_t stream);
template <typename T>
hipError_t launch_sp_gather_csc(const int* cq, const int* rowq, const T* csr_val, long nnz, int* csc_idx, T* csc_val, hipStream_t stream);
template <typename T>
hipError_t launch_sp_col_sumsq(const int* csc_ptr, const T* csc_val, int n, T* vtv, double* colsum, hipStream_t stream);
hipError_t launch_sp_boundaries(const int* ptr, const int* idx, int rows, long range, int blocks, int* bp, hipStream_t stream);

} // namespace nmfamd
