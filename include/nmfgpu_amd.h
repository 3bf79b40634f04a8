/* nmfgpu_amd.h -- C-ABI of the MI355X engine BELOW the nmfgpu.h boundary.
 *
 * nmfgpu.h (nmfgpu::compute and friends) is the drop-in boundary: host pointers in, host
 * pointers out, one blocking call per factorisation.  This header exposes the same engine
 * one level lower, for callers that want V to stay resident in HBM across calls (the benchmark
 * harness, the column-sharded multi-GPU driver, the per-kernel parity tests).  Plain C: opaque
 * handles, plain pointers and sizes, int status codes; no C++ or torch types.
 *
 * Each entry point names the reference code whose job it does (paths relative to the nmfgpu
 * v0.2.3 tree).  All matrices are column-major; "ld" is a leading dimension in elements.
 * The `_f32` / `_f64` suffix is the element type (the reference's NumericType = float / double).
 */
#ifndef NMFGPU_AMD_H
#define NMFGPU_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(NMFGPU_EXPORTING)
#  define NMFAMD_API __attribute__((visibility("default")))
#else
#  define NMFAMD_API
#endif

/* status codes */
enum {
	NMFAMD_OK = 0,
	NMFAMD_INVALID_ARGUMENT = 1,
	NMFAMD_NO_DEVICE_MEMORY = 2,
	NMFAMD_NO_HOST_MEMORY = 3,
	NMFAMD_HIP_ERROR = 4,
	NMFAMD_NO_DEVICE = 5,
	NMFAMD_VALUE_RANGE = 6   /* upload: V holds infinities, NaN, |v| > 2^126 or 0 < |v| < 2^-100, which the default fp32 product (operands split
	                            exactly into three bf16 terms) does not cover: create the engine with nmfamd_params.precision = -1 (native fp32
	                            MFMA instructions).  nmfgpu::compute and the Python Engine do that by themselves. */
};

/* algorithm ids = nmfgpu::NmfAlgorithm (include/nmfgpu.h:107-114) */
enum { NMFAMD_MU = 0, NMFAMD_GDCLS = 1, NMFAMD_ALS = 2, NMFAMD_ACLS = 3, NMFAMD_AHCLS = 4, NMFAMD_NSNMF = 5 };

/* sparse formats = nmfgpu::StorageFormat (include/nmfgpu.h:177-186) */
enum { NMFAMD_DENSE = 0, NMFAMD_CSR = 1, NMFAMD_CSC = 2, NMFAMD_COO = 3 };

/* Algorithm parameters, the reference's name/value list (Interface.cpp:41-49, 249-326) as a struct. */
typedef struct nmfamd_params {
	double lambda;   /* GDCLS  "lambda"  */
	double lambdaW;  /* ACLS / AHCLS "lambdaW" */
	double lambdaH;  /* ACLS / AHCLS "lambdaH" */
	double alphaW;   /* AHCLS "alphaW" */
	double alphaH;   /* AHCLS "alphaH" */
	double theta;    /* nsNMF "theta" */
	/* extensions without a reference counterpart (nmfgpu::compute: Parameter names "divergence", "sparseCompute") */
	double divergence;      /* 0 = Frobenius objective; 1 = generalised KL divergence (Multiplicative only; implies sparse_compute) */
	double sparse_compute;  /* 1 = keep V as CSR + CSC in HBM and use SpMM / SDDMM kernels instead of densifying (Multiplicative only) */
	double precision;       /* Parameter "precision", float engines only.  0 = fp32 accuracy: products on the bf16 matrix pipe with every
	                           operand split exactly into three bf16 terms (kernels_x3.hip); -1 = native fp32 MFMA instructions;
	                           1 = operands rounded to bf16 (reduced precision, half the bytes of V per pass) */
} nmfamd_params;

typedef struct nmfamd_engine nmfamd_engine;  /* opaque; owns every device buffer of one factorisation */

/* Number of visible HIP devices (0 when there is none), and the library's build description. */
NMFAMD_API int nmfamd_device_count(void);
NMFAMD_API const char* nmfamd_build_info(void);
/* Text of the last failed HIP call on this thread's engine (diagnostics only). */
NMFAMD_API const char* nmfamd_engine_last_error(const nmfamd_engine* e);

/* Creates the device state for V (m x n) ~ W (m x r) H (r x n) on the CURRENT HIP device.
 * Replaces IAlgorithm::allocateMemory (e.g. AlgorithmMultiplicativeFrobenius.h:85-128) minus the
 * upload.  elem_bytes: 4 (float) or 8 (double).  stream: a hipStream_t (0 = the null stream);
 * every kernel and copy of this engine is issued on it. */
NMFAMD_API int nmfamd_engine_create(int m, int n, int r, int algorithm, const nmfamd_params* params,
                                    int elem_bytes, void* stream, nmfamd_engine** out);
NMFAMD_API void nmfamd_engine_destroy(nmfamd_engine* e);

/* Upload V from host memory; builds V, its transpose and the sorted tr(V^T V) vector.
 * Replaces DeviceMatrix::copyFrom(inputMatrix) + the trace kernel + host sort
 * (AlgorithmMultiplicativeFrobenius.h:118-126; sparse -> dense with index base: Matrix.h:145-232). */
NMFAMD_API int nmfamd_engine_upload_dense(nmfamd_engine* e, const void* V, long ld);
/* format CSR: a = rowPtr (m+1), b = column indices; CSC: a = columnPtr (n+1), b = row indices;
 * COO: a = row indices, b = column indices (nnz each).  base = 0 or 1. */
NMFAMD_API int nmfamd_engine_upload_sparse(nmfamd_engine* e, int format, const void* values, const int* a, const int* b, long nnz, int base);

/* W: m x r, H: r x n, host memory; a null pointer leaves that factor untouched.
 * Replaces CopyStrategy (source/init/CopyStrategy.h:41-46) and storeFactorization. */
NMFAMD_API int nmfamd_engine_set_factors(nmfamd_engine* e, const void* W, long ldw, const void* H, long ldh);
NMFAMD_API int nmfamd_engine_get_factors(nmfamd_engine* e, void* W, long ldw, void* H, long ldh);
/* Uniform (0,1] fill, the same seed for both factors (source/init/RandomValueStrategy.cpp:29-70). */
NMFAMD_API int nmfamd_engine_randomize(nmfamd_engine* e, unsigned seed, int w, int h);

/* `count` iterations of IAlgorithm::computeIteration (Algorithm.h:59).  Iterations are numbered
 * first_iteration, first_iteration+1, ...; the error is evaluated when the number is a multiple of
 * error_every (10 in the reference, SingleGpuDispatcher.h:37; 0 = never) or equals last_iteration
 * (0 = no such iteration).  Returns after the work has been ENQUEUED; the error terms of an error
 * iteration travel to the host asynchronously and nmfamd_engine_frobenius / _rmsd wait for them
 * (nmfgpu::compute reads them after every error iteration, like the reference's dispatcher). */
NMFAMD_API int nmfamd_engine_iterate(nmfamd_engine* e, int count, int first_iteration, int error_every, int last_iteration, int constant_w);
NMFAMD_API int nmfamd_engine_synchronize(nmfamd_engine* e);
/* Frobenius norm / RMSD of the most recent error iteration (IAlgorithm::frobeniusNorm / rmsd). */
NMFAMD_API double nmfamd_engine_frobenius(nmfamd_engine* e);
NMFAMD_API double nmfamd_engine_rmsd(nmfamd_engine* e);
/* Generalised KL divergence D(V || W H) of the most recent error iteration (divergence = 1 only). */
NMFAMD_API double nmfamd_engine_kl_divergence(nmfamd_engine* e);

/* Dominant-kernel timing.  enable = k > 0: every launch of the factor-product kernel (the two
 * products against V, reference: gemm TN / NT at AlgorithmMultiplicativeFrobenius.h:187-188,240-241)
 * in every k-th iteration is bracketed by HIP events on the engine's stream (k = 1: all launches;
 * an event pair costs a few microseconds of stream time, so the harness samples).  enable = 0: off.
 * _read synchronises, returns the summed duration and the number of launches timed since the last
 * read, and resets the counters. */
NMFAMD_API int nmfamd_engine_kernel_timing(nmfamd_engine* e, int enable);
NMFAMD_API int nmfamd_engine_kernel_timing_read(nmfamd_engine* e, double* total_ms, long* launches);
/* The same, plus what an EMPTY event pair reports on the idle stream (ms): the share of each sample that is not kernel time. */
NMFAMD_API int nmfamd_engine_kernel_timing_read2(nmfamd_engine* e, double* total_ms, long* launches, double* pair_overhead_ms);
/* ... and split by product: kind_ms[2] / kind_launches[2] = {H-side product (W^T V; y-tiled form when one image of V is resident), W-side product (V H^T)} */
NMFAMD_API int nmfamd_engine_kernel_timing_read3(nmfamd_engine* e, double* total_ms, long* launches, double* pair_overhead_ms, double* kind_ms, long* kind_launches);

/* Geometry the harness needs for its roofline arithmetic. */
typedef struct nmfamd_geometry {
	int m, n, r;
	int padded_rank;       /* RP: factor rows as stored and multiplied */
	long padded_m, padded_n;
	int slabs_h, slabs_w;  /* split-K slices of the two factor products */
	long exchange_count;   /* elements of the multi-GPU exchange buffer */
	int product_kernel;    /* 0 fp32 MFMA, 1 bf16-rounded operands, 2 fp32 by exact 3 x bf16 operand splitting, 3 fp64 MFMA,
	                          4 VALU kernel (NMFAMD_FORCE_VALU), 5 sparse SpMM */
	int resident_images;   /* dense images of V kept in HBM: 2 (V and V^T, each streamed along its output index), 1 (only V: W^T V
	                          reads it along the reduction index; chosen when two would not fit, or by NMFAMD_ONE_IMAGE), 0 sparse */
	int one_pass;          /* rank-64 multiplicative update: 1 = V is streamed ONCE per iteration (W^T V, the H update and V H^T in one
	                          persistent launch, kernels_onepass.hip; the MEASUREMENT build only since round 6: NMFAMD_ONE_PASS=1 there); 0 = two passes; 2 = a one-pass launch
	                          gave up (it could not keep its workgroups resident) and the engine reported the error */
	/* round 5: the counts that fix the ORDER of partial sums (and so the bits of a result), so that a run can be reproduced: all of them functions of the
	 * shape and of the device's properties only (never of what else occupies the device) */
	int kl_blocks_w, kl_blocks_h;   /* KL update: L2-sized blocks the gathered factor is cut into (1 = unblocked; 0 = not the KL update) */
	int gram_k_slices;              /* rank-64 multiplicative update: K slices of the W^T W passengers (1, 2, 4, 8), or 16: the ten tiles' K ranges dealt evenly
	                                   to the sixteen passengers of a whole problem's W^T V launch (up to three pieces per tile, added in order) */
	int w_col_split;                /* 1: V H^T runs as 128 x 32 workgroups (narrow column shards) */
	/* round 6 */
	int fused_launches;             /* 4: an iteration is product (+ Gram passengers) / update / product (+ Gram passengers) / update -- fp32 at padded rank 64
	                                   (multiplicative update; nsNMF on the split-operand products) and, round 6, double precision (multiplicative update and nsNMF,
	                                   padded ranks up to 512); 8: fp32 at padded ranks 128 ... 512 (Gram slices, reduction + split image + scale, product, update --
	                                   twice; the generic sequence: 14), 7 or 6 where the Gram slices of a side ride its product launch; 0: the generic launch sequence */
	int sparse_setup;               /* sparse compute: where the CSR + CSC images of the last upload were built: 1 = on the device (kernels_sparse_setup.hip),
	                                   0 = on the host (NMFAMD_SPARSE_SETUP=host, or an input the device path hands back: entries outside the matrix, a row or
	                                   column of more than 8 192 entries, pointer arrays that do not ascend), -1 = not a sparse-compute engine */
	int gram_ride_slices_h, gram_ride_slices_w;   /* K slices per super-block of the Gram passengers riding in W^T V / V H^T -- double precision (64 x 64 super-blocks) and
	                                                 fp32 at padded ranks 128 ... 512 (128 x 128; 0: the slices are a launch of their own); they fix the order of the partial sums */
} nmfamd_geometry;
NMFAMD_API int nmfamd_engine_geometry(const nmfamd_engine* e, nmfamd_geometry* out);
/* The same for a caller compiled against an older (shorter) or newer (longer) nmfamd_geometry: writes min(struct_size, sizeof(nmfamd_geometry)) bytes, never
 * past the caller's struct (the struct only ever grows at its end; round 3 added `one_pass`, round 5 the four counts behind it, round 6 three more).  Returns NMFAMD_INVALID_ARGUMENT for struct_size < 8. */
NMFAMD_API int nmfamd_engine_geometry_sized(const nmfamd_engine* e, void* out, unsigned long struct_size);

/* ---- column-sharded multi-GPU form of the multiplicative update ------------------------------
 * Rank g holds V(:, J_g), H(:, J_g) and a full replica of W.  Per iteration:
 *     nmfamd_engine_h_step        local: H(:, J_g) update, no communication
 *     nmfamd_engine_w_products    local: exchange <- [ (V_g H_g^T)^T panel | H_g H_g^T ]
 *     <all-reduce(sum) of `exchange` across ranks -- RCCL, done by the caller>
 *     nmfamd_engine_w_finish      replicated: W update + column normalisation from the reduced sums
 * `exchange` is a DEVICE pointer to exchange_count elements owned by the caller (so that the
 * collective library can register it).  On error iterations h_step / w_finish leave the local
 * error terms on the host: n_local per-column terms of tr(H^T W^T V), r terms of
 * tr(H H^T W^T W) computed from the REDUCED H H^T, and the sorted local tr(V^T V) terms. */
NMFAMD_API int nmfamd_engine_h_step(nmfamd_engine* e, int compute_error);
NMFAMD_API int nmfamd_engine_w_products(nmfamd_engine* e, void* exchange);
NMFAMD_API int nmfamd_engine_w_finish(nmfamd_engine* e, const void* exchange, int compute_error);
/* A caller that drives the three phases for a team of ONE rank (no reduction between w_products and w_finish: `exchange` reaches w_finish as w_products
 * left it) may say so: the engine then skips work that only a reduced buffer needs (rank 256 / bf16: re-splitting H H^T).  sole != 0 is a promise. */
NMFAMD_API int nmfamd_engine_set_sole_rank(nmfamd_engine* e, int sole);
/* Row-block form of nmfamd_engine_w_finish for callers that reduce-scatter the panel themselves (the torch-facing wrapper
 * nmfgpu_amd/distributed.py; the native loop nmfamd_sharded_* does the same inside the library): num_rows = the reduced
 * (V H^T)^T rows [row0, row0 + rows) in panel layout (DEVICE), hht = the reduced H H^T (DEVICE, padded_rank^2), colsq =
 * padded_rank DEVICE elements that receive the sums of squares of the new rows.  After the all-reduce of colsq:
 * _w_normalize_rows; after the all-gather of every rank's rows into nmfamd_engine_w_panel(): _w_rows_replaced.
 * rows and row0 are multiples of 128 (engine created with nmfamd_engine_create_blocks). */
NMFAMD_API int nmfamd_engine_w_update_rows(nmfamd_engine* e, const void* num_rows, const void* hht, long row0, long rows, int compute_error, void* colsq);
NMFAMD_API int nmfamd_engine_w_normalize_rows(nmfamd_engine* e, long row0, long rows, void* colsq);
NMFAMD_API int nmfamd_engine_w_rows_replaced(nmfamd_engine* e);
NMFAMD_API void* nmfamd_engine_w_panel(nmfamd_engine* e);     /* DEVICE pointer: padded_m rows of padded_rank elements */
/* which: 0 = sorted tr(V^T V) terms (n), 1 = tr(H^T W^T V) terms (n), 2 = tr(H H^T W^T W) terms (r).
 * Returns the number of elements copied (<= capacity), negative on error. */
NMFAMD_API long nmfamd_engine_error_terms(nmfamd_engine* e, int which, void* out, long capacity);
/* Stream-ordered form for callers that must not stall the stream on error iterations: copies the n_local
 * tr(H^T W^T V) terms followed by the r tr(H H^T W^T W) terms of the last error iteration into the DEVICE buffer
 * `dst` (device-to-device, asynchronous on the engine's stream).  Returns n_local + r, negative on error. */
NMFAMD_API long nmfamd_engine_error_terms_to_device(nmfamd_engine* e, void* dst_device, long capacity);
/* The host half of the error evaluation (source/nmf/FrobeniusResolver.cpp:29-51) on caller-supplied
 * term vectors (the two latter ones are sorted in place). */
NMFAMD_API double nmfamd_resolve_frobenius_f32(const float* vtv_sorted, long n_vtv, float* htwtv, long n_htwtv, float* hhtwtw, long n_hhtwtw);
NMFAMD_API double nmfamd_resolve_frobenius_f64(const double* vtv_sorted, long n_vtv, double* htwtv, long n_htwtv, double* hhtwtw, long n_hhtwtw);

/* ---- the same sharded iteration driven natively (no torch in the loop) -------------------------------------------
 * A communicator is one rank of an RCCL clique, created through RCCL's C API (librccl.so is loaded on first use):
 * rank 0 obtains 128 bytes of unique id, hands them to the other ranks by whatever means the caller has (bench.py:
 * torch.distributed's store; nmfgpu::compute with Parameter "numGpus": threads of one process share it directly), and every
 * rank calls nmfamd_comm_create_rccl with ITS HIP device current -- the call blocks until all `world` ranks have arrived.
 * A sharded run binds one engine (created with nmfamd_engine_create_blocks(..., row_blocks = world)) to one communicator:
 *   mode 0  reduce-scatter of (V H^T)^T by row blocks of W + all-reduce of H H^T -> every rank updates its m / world rows
 *           -> all-reduce of the r column sums of squares -> normalise -> all-gather of the row blocks   
 *   mode 1  one all-reduce of the whole exchange buffer, identical W update on every rank
 * rows / total_columns: shape of the WHOLE matrix; this rank's engine holds the columns
 * [total_columns * rank / world, total_columns * (rank + 1) / world).  nmfamd_sharded_iterate numbers iterations like
 * nmfamd_engine_iterate and only ENQUEUES work; _frobenius / _rmsd wait for the gathered error terms of the most recent
 * error iteration and run the reference's sorted host summation on them (FrobeniusResolver.cpp:29-51). */
typedef struct nmfamd_comm nmfamd_comm;
typedef struct nmfamd_sharded nmfamd_sharded;
NMFAMD_API int nmfamd_comm_rccl_available(void);
NMFAMD_API int nmfamd_comm_unique_id(void* out_128_bytes);
NMFAMD_API int nmfamd_comm_create_rccl(const void* id_128_bytes, int world, int rank, nmfamd_comm** out);
NMFAMD_API void nmfamd_comm_destroy(nmfamd_comm* c);
/* The in-process transport (csrc/comm.h, LocalComm): the ranks are THREADS of one process, each with its own HIP device current (the same device, or
 * peer-mapped devices of one node); every rank's kernels read the peers' buffers where they lie -- over xGMI when the devices differ.  This is the
 * transport of nmfgpu::compute with Parameter "numGpus" when RCCL is not used, and the one whose multiplicative update needs no reduction kernel at
 * all (replicated mode at padded rank 64: the W update adds the ranks' exchange panels in its prologue, two buffers alternate, one rendezvous per
 * iteration).  One thread creates the group, every rank thread calls nmfamd_comm_create_local (blocks until all `world` ranks have joined; fails on
 * every rank when two devices cannot map each other's memory).  nmfamd_local_group_abort makes every rank waiting in a collective return an error. */
typedef struct nmfamd_local_group nmfamd_local_group;
NMFAMD_API int nmfamd_local_group_create(int world, nmfamd_local_group** out);
NMFAMD_API void nmfamd_local_group_destroy(nmfamd_local_group* g);
NMFAMD_API void nmfamd_local_group_abort(nmfamd_local_group* g);
NMFAMD_API int nmfamd_comm_create_local(nmfamd_local_group* g, int rank, nmfamd_comm** out);
/* why nmfamd_comm_create_local failed: names the pair of devices that cannot map each other's memory (valid until the calling thread's next call) */
NMFAMD_API const char* nmfamd_local_group_last_error(nmfamd_local_group* g);
/* Set-up self-test of the in-process transport (run by nmfamd_comm_create_local when world > 1; NMFAMD_SELFTEST=0 skips it): every rank publishes a pattern in
 * both exchange slots and in a collective's buffer, every rank reads every peer's through the kernels the iteration uses (the W update's prologue, the r x r
 * sum, the reduce / gather kernels) and checks every word; then a second pattern at the SAME addresses.  A mismatch fails nmfamd_comm_create_local on every
 * rank and names the (reader, owner) pair in nmfamd_local_group_last_error.  One line about the test (empty before it ran; valid until the thread's next call).
 * Lifetime rules of the transport: the exchange buffers belong to the GROUP and live until every rank has closed its communicator; closing a communicator
 * (nmfamd_comm_destroy) first drains the rank's device, so that no kernel of this rank still reads a peer's buffer. */
NMFAMD_API const char* nmfamd_local_group_selftest(nmfamd_local_group* g);
/* "rccl" / "in-process (peer reads)" */
NMFAMD_API const char* nmfamd_comm_transport(const nmfamd_comm* c);
/* nmfamd_engine_create with the padded row count rounded up to a multiple of 128 * row_blocks (equal row blocks of W) */
NMFAMD_API int nmfamd_engine_create_blocks(int m, int n, int r, int algorithm, const nmfamd_params* params, int elem_bytes, void* stream,
                                           int row_blocks, nmfamd_engine** out);
NMFAMD_API int nmfamd_sharded_create(nmfamd_engine* e, nmfamd_comm* c, int mode, long rows, long total_columns, nmfamd_sharded** out);
NMFAMD_API void nmfamd_sharded_destroy(nmfamd_sharded* s);
NMFAMD_API int nmfamd_sharded_iterate(nmfamd_sharded* s, int count, int first_iteration, int error_every, int last_iteration);
/* Row-block mode at padded rank 256 with bf16 operands (config 4): between two W updates the ranks exchange the bf16 fragments of their row blocks only, so
 * the fp32 rows of the OTHER ranks' blocks in nmfamd_engine_w_panel() are those of an earlier iteration until they are gathered.  They are gathered
 *   (a) by nmfamd_sharded_iterate itself when a batch ends on its last_iteration (> 0), and
 *   (b) by this call -- a COLLECTIVE: every rank of the communicator must make it, at the same point of its stream.
 * nmfamd_engine_get_factors on an engine whose rows are still stale gathers them through the live sharded run (then it, too, is a collective that every rank must
 * make); after nmfamd_sharded_destroy it fails with NMFAMD_INVALID_ARGUMENT rather than hand out stale rows.  A no-op in every other mode. */
NMFAMD_API int nmfamd_sharded_gather_w(nmfamd_sharded* s);
NMFAMD_API double nmfamd_sharded_frobenius(nmfamd_sharded* s);
NMFAMD_API double nmfamd_sharded_rmsd(nmfamd_sharded* s);
NMFAMD_API const char* nmfamd_sharded_last_error(const nmfamd_sharded* s);

/* The host-side initialisers (run once per run, before the iteration loop; host memory only, no device or
 * context needed).  k-means: Lloyd with a Forgy start (source/kmeans/kMeans.cu:126-278); data is m x n with
 * leading dimension ld, clusters m x k (ldc), membership n entries; *iterations receives the passes done.
 * host_init: method is the NmfInitializationMethod value (include/nmfgpu.h:87-96) for MeanColumns,
 * KMeans* and EInNMF (source/init/KMeansStrategy.cpp:31-65, EInNMF.cu:44-119); W is m x r (ld m), H r x n
 * (ld r) or NULL.  method 100 / 101 / 102: the SVD-based start NNDSVD / NNDSVDa / NNDSVDar (Boutsidis & Gallopoulos 2008; not in the reference -- BASELINE's
 * north star names it -- nmfgpu::compute selects it with Parameter{"nndsvd", 0 | 1 | 2}, which overrides initMethod): truncated SVD of V by block subspace
 * iteration on the host, in double.  Return 0, or 1 (invalid argument) for arguments the reference rejects. */
NMFAMD_API int nmfamd_host_kmeans_f32(const float* data, long ld, int m, int n, float* clusters, long ldc, int k,
                                      unsigned* membership, unsigned seed, unsigned maxiter, double threshold, unsigned* iterations);
NMFAMD_API int nmfamd_host_kmeans_f64(const double* data, long ld, int m, int n, double* clusters, long ldc, int k,
                                      unsigned* membership, unsigned seed, unsigned maxiter, double threshold, unsigned* iterations);
NMFAMD_API int nmfamd_host_init_f32(const float* V, long ldv, int m, int n, int r, int method, unsigned seed, float* W, float* H);
NMFAMD_API int nmfamd_host_init_f64(const double* V, long ldv, int m, int n, int r, int method, unsigned seed, double* W, double* H);

/* ---- single operations on host data (parity tests of the individual kernels) -----------------
 * OUT (r x X) = F (r x Y) * A^T, A is X x Y: the factor product both big GEMMs are instances of.
 * out_slabs (optional) receives the number of split-K slabs the MFMA kernel used.
 * use_valu != 0 selects the generic VALU kernel (the cross-check) instead of the fp32 / fp64 MFMA kernel. */
NMFAMD_API int nmfamd_op_factor_product_f32(const float* A, long lda, int X, int Y, const float* F, long ldf, int r,
                                            float* OUT, long ldo, int use_valu, int* out_slabs);
NMFAMD_API int nmfamd_op_factor_product_f64(const double* A, long lda, int X, int Y, const double* F, long ldf, int r,
                                            double* OUT, long ldo, int use_valu, int* out_slabs);
/* Tuning / diagnosis of the factor-product kernel (rank 64) on synthetic device data: average of
 * `reps` back-to-back launches; optionally (stamps_out != NULL) one launch of the diagnostic build that
 * records 8 uint64 per wave: shader clock at entry / first MFMA / loop end / kernel end, 100 MHz
 * real-time counter at entry / end, K-steps, XCC id. */
NMFAMD_API int nmfamd_tune_factor_product(int X, int Y, int reps, double* avg_us, unsigned long long* stamps_out, long stamps_capacity, long* stamps_count);
/* The same product with bf16-rounded operands (v_mfma_f32_32x32x16_bf16, fp32 accumulation), r <= 64. */
NMFAMD_API int nmfamd_op_factor_product_bf16(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo);
/* The same product at fp32 accuracy on the bf16 matrix pipe: both operands split exactly into three bf16 terms, six
 * cross products, fp32 accumulation (kernels_x3.hip), any r.  reps > 0 additionally times `reps` launches of the
 * product kernel alone (HIP events) into *avg_us. */
NMFAMD_API int nmfamd_op_factor_product_x3(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo,
                                           int reps, double* avg_us);
/* The same with A stored tiled along the REDUCTION index y (the image the other product of an iteration streams along its
 * output index): one resident image of V serves both products. */
NMFAMD_API int nmfamd_op_factor_product_x3_ytiled(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo,
                                                  int reps, double* avg_us);
/* Diagnostic (NMFAMD_X3_VARIANT = 10..13 builds): per wave {shader cycles, 100 MHz ticks, K-steps of the main loop; 100 MHz
 * stamps at entry, loop start, loop end, tail end, exit} of one more launch; stamps_capacity in 8-byte words; *waves receives the number of waves stamped. */
NMFAMD_API int nmfamd_tune_factor_product_x3(int X, int Y, unsigned long long* stamps_out, long stamps_capacity, long* waves);
/* G (r x r) = P P^T for a host r x len matrix P. */
NMFAMD_API int nmfamd_op_gram_f32(const float* P, long ldp, int r, int len, float* G, long ldg);
NMFAMD_API int nmfamd_op_gram_f64(const double* P, long ldp, int r, int len, double* G, long ldg);
/* Ainv = (A + regulariser)^-1 for a host r x r matrix (offdiag / diag added as KernelFillMatrix.cu:29-45). */
NMFAMD_API int nmfamd_op_inverse_f32(const float* A, long lda, int r, float offdiag, float diag, float* Ainv, long ldi);
/* The passes between the update of a factor panel and the next product at padded rank 256 with bf16 product operands (129 <= r <= 256):
 * optional column normalisation (colsq: r sums of squares or NULL), nsNMF smoothing by theta, bf16 rounding; the Gram matrix of the panel and of
 * the smoothed panel.  P, P_out, pack_out: [len][ldp / r] rows of r values; G_raw, G_smooth: [r][r].  Outputs may be NULL. */
NMFAMD_API int nmfamd_op_factor_passes_f32(const float* P, long ldp, int r, int len, const float* colsq, float theta, float* P_out, float* pack_out,
                                           float* G_raw, float* G_smooth, int reps, double* avg_us_finish, double* avg_us_gram);
/* One multiplicative update of a panel (129 <= r <= 256) as the bf16 path at padded rank 256 runs it (csrc/kernels.h, PanelTriExtras).  A pending column
 * scale is given as the sums of squares it comes from: d(c) = s[c] > 0 ? 1 / sqrt(s[c]) : 1; S = (1 - theta) I + (theta / r) 1 1^T.
 *   old'(y, c) = P(y, c) * d_old(c)                                       (old_colsq NULL: ones)
 *   num'(y, :) = S (d_num .* num(y, :))                                    (transform_num != 0; num_colsq NULL: ones)
 *   den(y, :)  = old'(y, :) Q + eps,  or  S D_num Q D_num S old'(y, :) + eps when transform_den != 0
 *   P_out = old' .* num' ./ den                                            (unnormalised)
 *   pack_out = bf16 of P_out as written for the next product, smoothed by frag_theta first when that is not 0
 *   scale_out(c) = d of the columns of P_out, gram_out = diag(scale) pack^T pack diag(scale);
 *   gram_raw_out = pack^T pack, gram_image_out = the same matrix read back from the split image the reduction writes beside it, diag_out = its diagonal.
 * Row-major [len][r] and [r][r]; outputs may be NULL. */
NMFAMD_API int nmfamd_op_tri_update_f32(const float* P, const float* num, const float* Q, int r, int len, const float* old_colsq, int transform_num,
                                        const float* num_colsq, float theta, float frag_theta, int transform_den, float* P_out, float* pack_out,
                                        float* scale_out, float* gram_out, float* gram_raw_out, float* gram_image_out, float* diag_out);
/* Test access to an engine's device intermediates in panel layout: which = 0 Wt, 1 H, 2 W^T W,
 * 3 H H^T, 4 slabs, 5 inverse, 6 V, 7 Vt; rank-256 fp32 engines also 8 W^T W as last reduced, 9 staged column sums of squares,
 * 10 / 11 the bf16 fragments of W / H as 4-byte words. */
NMFAMD_API int nmfamd_engine_debug_read(nmfamd_engine* e, int which, void* out, long count);

#ifdef __cplusplus
}
#endif
#endif /* NMFGPU_AMD_H */
