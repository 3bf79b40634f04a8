// amd_api.cpp -- the extern "C" entry points of include/nmfgpu_amd.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <memory>
#include <new>
#include <vector>

#include "../../include/nmfgpu_amd.h"
#include "comm.h"
#include "engine.h"
#include "tuning.h"
#include "host_init.h"
#include "sharded.h"

using namespace nmfamd;

struct nmfamd_engine {
	int elem_bytes;
	std::unique_ptr<Engine<float>> f;
	std::unique_ptr<Engine<double>> d;
};

struct nmfamd_comm {
	std::unique_ptr<Comm> c;
};

struct nmfamd_local_group {
	std::shared_ptr<LocalGroup> g;
};

struct nmfamd_sharded {
	int elem_bytes;
	std::unique_ptr<ShardedRank<float>> f;
	std::unique_ptr<ShardedRank<double>> d;
};

namespace {
template <typename Fn32, typename Fn64>
int dispatch(nmfamd_engine* e, Fn32 f32, Fn64 f64) {
	if (!e) return NMFAMD_INVALID_ARGUMENT;
	return e->elem_bytes == 4 ? (int)f32(*e->f) : (int)f64(*e->d);
}
}

namespace {
struct DevBuf {
	void* p = nullptr;
	~DevBuf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t bytes) { hipError_t e = hipMalloc(&p, bytes ? bytes : 16); if (e == hipSuccess) e = hipMemset(p, 0, bytes ? bytes : 16); return e; }
};

template <typename T>
int op_factor_product(const T* A, long lda, int X, int Y, const T* F, long ldf, int r, T* OUT, long ldo, bool use_valu, int* out_slabs) {
	if (!A || !F || !OUT || X <= 0 || Y <= 0 || r <= 0 || lda < X || ldf < r || ldo < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r, sizeof(T));
	int dev = 0; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMFAMD_HIP_ERROR;
	FactorProductPlan plan = std::is_same<T, double>::value ? plan_factor_product_f64(X, Y, RP, prop.multiProcessorCount)
	                                                         : plan_factor_product(X, Y, RP, prop.multiProcessorCount);
	if (RP == 64 && r <= 32) plan.nb = std::is_same<T, float>::value ? 1 : 2;      // as the engine does: 32 panel columns for small ranks (fp64 counts 16-column tiles)
	const long Xp = pad128(std::max<long>(X, (long)plan.xtiles * plan.th)), Yp = pad128(Y);
	const bool mfma = !use_valu;
	const int S = mfma ? plan.splits : 1;
	DevBuf dA, dF, dS, dO;
	const long slab_stride = (long)RP * Xp;
	if (dA.alloc(sizeof(T) * Xp * Yp) != hipSuccess || dF.alloc(sizeof(T) * RP * Yp) != hipSuccess ||
	    dS.alloc(sizeof(T) * slab_stride * S) != hipSuccess || dO.alloc(sizeof(T) * slab_stride) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemcpy2D(dA.p, Xp * sizeof(T), A, lda * sizeof(T), X * sizeof(T), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dF.p, RP * sizeof(T), F, ldf * sizeof(T), r * sizeof(T), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	hipError_t e;
	if constexpr (std::is_same<T, float>::value) {
		if (mfma) {
			DevBuf dT;   // x-tiled image of A for the MFMA kernel
			if (dT.alloc(sizeof(T) * Xp * Yp) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
			e = launch_tile<float>((const float*)dA.p, Xp, X, Y, (float*)dT.p, plan.th * Yp, plan.th, false, nullptr);
			if (e == hipSuccess) e = launch_factor_product_f32(plan, (const float*)dT.p, plan.th * Yp, (const float*)dF.p, RP, (float*)dS.p, slab_stride, nullptr);
			if (e == hipSuccess) e = hipDeviceSynchronize();
		}
		else e = launch_factor_product_valu<T>((const T*)dA.p, Xp, (int)Xp, Y, (const T*)dF.p, RP, (T*)dS.p, nullptr);
	} else {
		if (mfma) {
			DevBuf dT;   // x-tiled image of A for the fp64 MFMA kernel
			if (dT.alloc(sizeof(T) * Xp * Yp) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
			e = hipMemset(dT.p, 0, sizeof(T) * Xp * Yp);
			if (e == hipSuccess) e = launch_tile<double>((const double*)dA.p, Xp, X, Y, (double*)dT.p, plan.th * Yp, plan.th, false, nullptr);
			if (e == hipSuccess) e = launch_factor_product_f64(plan, (const double*)dT.p, plan.th * Yp, (const double*)dF.p, RP, (double*)dS.p, slab_stride, nullptr);
			if (e == hipSuccess) e = hipDeviceSynchronize();
		}
		else e = launch_factor_product_valu<T>((const T*)dA.p, Xp, (int)Xp, Y, (const T*)dF.p, RP, (T*)dS.p, nullptr);
	}
	if (e != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_reduce_slabs<T>((const T*)dS.p, S, slab_stride, (T*)dO.p, slab_stride, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(OUT, ldo * sizeof(T), dO.p, RP * sizeof(T), r * sizeof(T), X, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipDeviceSynchronize() != hipSuccess) return NMFAMD_HIP_ERROR;
	if (out_slabs) *out_slabs = S;
	return NMFAMD_OK;
}
}

namespace {
template <typename T>
int host_kmeans(const T* data, long ld, int m, int n, T* clusters, long ldc, int k, unsigned* membership, unsigned seed,
                unsigned maxiter, double threshold, unsigned* iterations) {
	if (!data || !clusters || m <= 0 || n <= 0 || k <= 0 || ld < m || ldc < m) return nmfamd::ST_INVALID;
	nmfgpu::KMeansDescription<T> d{};
	d.inputMatrix.rows = (unsigned)m; d.inputMatrix.columns = (unsigned)n; d.inputMatrix.format = nmfgpu::StorageFormat::Dense;
	d.inputMatrix.dense.values = const_cast<T*>(data); d.inputMatrix.dense.leadingDimension = (unsigned)ld;
	d.outputMatrixClusters.rows = (unsigned)m; d.outputMatrixClusters.columns = (unsigned)k; d.outputMatrixClusters.format = nmfgpu::StorageFormat::Dense;
	d.outputMatrixClusters.dense.values = clusters; d.outputMatrixClusters.dense.leadingDimension = (unsigned)ldc;
	d.outputMemberships = membership; d.numClusters = (unsigned)k; d.numIterations = maxiter; d.seed = seed; d.thresholdValue = threshold;
	nmfgpu::KMeansSummary s{};
	if (nmfgpu::hostinit::compute_kmeans<T>(d, &s) != nmfgpu::ResultType::Success) return nmfamd::ST_INVALID;
	if (iterations) *iterations = s.iterations;
	return nmfamd::ST_OK;
}

template <typename T>
int host_init(const T* V, long ldv, int m, int n, int r, int method, unsigned seed, T* W, T* H) {
	if (!V || !W || m <= 0 || n <= 0 || r <= 0 || ldv < m) return nmfamd::ST_INVALID;
	nmfgpu::NmfDescription<T> d{};
	d.inputMatrix.rows = (unsigned)m; d.inputMatrix.columns = (unsigned)n; d.inputMatrix.format = nmfgpu::StorageFormat::Dense;
	d.inputMatrix.dense.values = const_cast<T*>(V); d.inputMatrix.dense.leadingDimension = (unsigned)ldv;
	d.features = (unsigned)r; d.seed = seed;
	// method 100 + v: the SVD-based start (NNDSVD, v = 0 plain / 1 "a" / 2 "ar"), which nmfgpu::compute selects with Parameter{"nndsvd", v}
	nmfgpu::Parameter svd{"nndsvd", (double)(method - 100)};
	if (method >= 100 && method <= 102) {
		if (r > std::min(m, n)) return nmfamd::ST_INVALID;
		d.initMethod = nmfgpu::NmfInitializationMethod::AllRandomValues; d.parameters = &svd; d.numParameters = 1;
		return nmfgpu::hostinit::initialize<T>(d, W, H) ? nmfamd::ST_OK : nmfamd::ST_INVALID;
	}
	d.initMethod = (nmfgpu::NmfInitializationMethod)method;
	if ((d.initMethod != nmfgpu::NmfInitializationMethod::MeanColumns) && r >= n) return nmfamd::ST_INVALID;
	return nmfgpu::hostinit::initialize<T>(d, W, H) ? nmfamd::ST_OK : nmfamd::ST_INVALID;
}
} // namespace

extern "C" {

int nmfamd_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

const char* nmfamd_build_info(void) {
	return "nmfgpu-amd 0.2.3 (gfx950 HIP kernels: fp32 MFMA factor product, fused panel updates; no vendor BLAS)";
}

const char* nmfamd_engine_last_error(const nmfamd_engine* e) {
	if (!e) return "";
	return e->elem_bytes == 4 ? e->f->last_error() : e->d->last_error();
}

int nmfamd_engine_create(int m, int n, int r, int algorithm, const nmfamd_params* params, int elem_bytes, void* stream, nmfamd_engine** out) {
	return nmfamd_engine_create_blocks(m, n, r, algorithm, params, elem_bytes, stream, 1, out);
}

int nmfamd_engine_create_blocks(int m, int n, int r, int algorithm, const nmfamd_params* params, int elem_bytes, void* stream, int row_blocks, nmfamd_engine** out) {
	if (!out || (elem_bytes != 4 && elem_bytes != 8) || row_blocks < 1 || row_blocks > 64) return NMFAMD_INVALID_ARGUMENT;
	*out = nullptr;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	AlgorithmParams p;
	if (params) { p.lambda = params->lambda; p.lambdaW = params->lambdaW; p.lambdaH = params->lambdaH; p.alphaW = params->alphaW; p.alphaH = params->alphaH; p.theta = params->theta; p.divergence = params->divergence; p.sparse_compute = params->sparse_compute; p.precision = params->precision; }
	nmfamd_engine* e = new (std::nothrow) nmfamd_engine();
	if (!e) return NMFAMD_NO_HOST_MEMORY;
	e->elem_bytes = elem_bytes;
	Status st;
	try {
		if (elem_bytes == 4) { e->f.reset(new Engine<float>(m, n, r, algorithm, p)); e->f->set_stream((hipStream_t)stream); e->f->set_row_blocks(row_blocks); st = e->f->allocate(); }
		else { e->d.reset(new Engine<double>(m, n, r, algorithm, p)); e->d->set_stream((hipStream_t)stream); e->d->set_row_blocks(row_blocks); st = e->d->allocate(); }
	} catch (const std::bad_alloc&) { delete e; return NMFAMD_NO_HOST_MEMORY; }
	if (st != ST_OK) { delete e; return (int)st; }
	*out = e;
	return NMFAMD_OK;
}

void nmfamd_engine_destroy(nmfamd_engine* e) { delete e; }

int nmfamd_comm_rccl_available(void) { return rccl_available() ? 1 : 0; }

int nmfamd_comm_unique_id(void* out_128_bytes) { return (int)rccl_unique_id(out_128_bytes); }

int nmfamd_comm_create_rccl(const void* id_128_bytes, int world, int rank, nmfamd_comm** out) {
	if (!out) return NMFAMD_INVALID_ARGUMENT;
	*out = nullptr;
	nmfamd_comm* c = new (std::nothrow) nmfamd_comm();
	if (!c) return NMFAMD_NO_HOST_MEMORY;
	Status st = rccl_comm_create(id_128_bytes, world, rank, &c->c);
	if (st != ST_OK) { delete c; return (int)st; }
	*out = c;
	return NMFAMD_OK;
}

void nmfamd_comm_destroy(nmfamd_comm* c) { delete c; }

int nmfamd_local_group_create(int world, nmfamd_local_group** out) {
	if (!out) return NMFAMD_INVALID_ARGUMENT;
	*out = nullptr;
	std::shared_ptr<LocalGroup> g = local_group_create(world);
	if (!g) return NMFAMD_INVALID_ARGUMENT;
	nmfamd_local_group* h = new (std::nothrow) nmfamd_local_group();
	if (!h) return NMFAMD_NO_HOST_MEMORY;
	h->g = std::move(g);
	*out = h;
	return NMFAMD_OK;
}

void nmfamd_local_group_destroy(nmfamd_local_group* g) { delete g; }
void nmfamd_local_group_abort(nmfamd_local_group* g) { if (g && g->g) local_group_abort(*g->g); }

int nmfamd_comm_create_local(nmfamd_local_group* g, int rank, nmfamd_comm** out) {
	if (!g || !g->g || !out) return NMFAMD_INVALID_ARGUMENT;
	*out = nullptr;
	nmfamd_comm* c = new (std::nothrow) nmfamd_comm();
	if (!c) { local_group_abort(*g->g); return NMFAMD_NO_HOST_MEMORY; }
	Status st = local_comm_create(g->g, rank, &c->c);
	if (st != ST_OK) { delete c; return (int)st; }
	*out = c;
	return NMFAMD_OK;
}

const char* nmfamd_local_group_last_error(nmfamd_local_group* g) {
	static thread_local std::string text;
	text = (g && g->g) ? local_group_failure(*g->g) : std::string();
	return text.c_str();
}

const char* nmfamd_local_group_selftest(nmfamd_local_group* g) {
	static thread_local std::string text;
	text = (g && g->g) ? local_group_selftest(*g->g) : std::string();
	return text.c_str();
}

const char* nmfamd_comm_transport(const nmfamd_comm* c) { return (c && c->c) ? c->c->transport() : ""; }

int nmfamd_sharded_create(nmfamd_engine* e, nmfamd_comm* c, int mode, long rows, long total_columns, nmfamd_sharded** out) {
	if (!e || !c || !c->c || !out) return NMFAMD_INVALID_ARGUMENT;
	*out = nullptr;
	nmfamd_sharded* s = new (std::nothrow) nmfamd_sharded();
	if (!s) return NMFAMD_NO_HOST_MEMORY;
	s->elem_bytes = e->elem_bytes;
	Status st;
	if (e->elem_bytes == 4) { s->f.reset(new ShardedRank<float>(e->f.get(), c->c.get(), mode, rows, total_columns)); st = s->f->prepare(); }
	else { s->d.reset(new ShardedRank<double>(e->d.get(), c->c.get(), mode, rows, total_columns)); st = s->d->prepare(); }
	if (st != ST_OK) { delete s; return (int)st; }
	*out = s;
	return NMFAMD_OK;
}

void nmfamd_sharded_destroy(nmfamd_sharded* s) { delete s; }

int nmfamd_sharded_iterate(nmfamd_sharded* s, int count, int first_iteration, int error_every, int last_iteration) {
	if (!s) return NMFAMD_INVALID_ARGUMENT;
	return (int)(s->elem_bytes == 4 ? s->f->run(count, first_iteration, error_every, last_iteration) : s->d->run(count, first_iteration, error_every, last_iteration));
}

int nmfamd_sharded_gather_w(nmfamd_sharded* s) {
	if (!s) return NMFAMD_INVALID_ARGUMENT;
	return (int)(s->elem_bytes == 4 ? s->f->gather_w_rows() : s->d->gather_w_rows());
}

double nmfamd_sharded_frobenius(nmfamd_sharded* s) { return !s ? 0.0 : (s->elem_bytes == 4 ? s->f->frobenius() : s->d->frobenius()); }
double nmfamd_sharded_rmsd(nmfamd_sharded* s) { return !s ? 0.0 : (s->elem_bytes == 4 ? s->f->rmsd() : s->d->rmsd()); }
const char* nmfamd_sharded_last_error(const nmfamd_sharded* s) { return !s ? "" : (s->elem_bytes == 4 ? s->f->last_error() : s->d->last_error()); }

int nmfamd_engine_upload_dense(nmfamd_engine* e, const void* V, long ld) {
	return dispatch(e, [&](Engine<float>& g) { return g.upload_dense((const float*)V, ld); },
	                   [&](Engine<double>& g) { return g.upload_dense((const double*)V, ld); });
}

int nmfamd_engine_upload_sparse(nmfamd_engine* e, int format, const void* values, const int* a, const int* b, long nnz, int base) {
	return dispatch(e, [&](Engine<float>& g) { return g.upload_sparse(format, (const float*)values, a, b, nnz, base); },
	                   [&](Engine<double>& g) { return g.upload_sparse(format, (const double*)values, a, b, nnz, base); });
}

int nmfamd_engine_set_factors(nmfamd_engine* e, const void* W, long ldw, const void* H, long ldh) {
	return dispatch(e, [&](Engine<float>& g) { return g.set_factors((const float*)W, ldw, (const float*)H, ldh); },
	                   [&](Engine<double>& g) { return g.set_factors((const double*)W, ldw, (const double*)H, ldh); });
}

int nmfamd_engine_get_factors(nmfamd_engine* e, void* W, long ldw, void* H, long ldh) {
	return dispatch(e, [&](Engine<float>& g) { return g.get_factors((float*)W, ldw, (float*)H, ldh); },
	                   [&](Engine<double>& g) { return g.get_factors((double*)W, ldw, (double*)H, ldh); });
}

int nmfamd_engine_randomize(nmfamd_engine* e, unsigned seed, int w, int h) {
	return dispatch(e, [&](Engine<float>& g) { return g.randomize_factors(seed, w != 0, h != 0); },
	                   [&](Engine<double>& g) { return g.randomize_factors(seed, w != 0, h != 0); });
}

int nmfamd_engine_iterate(nmfamd_engine* e, int count, int first_iteration, int error_every, int last_iteration, int constant_w) {
	auto run = [&](auto& g) -> Status {
		for (int k = 0; k < count; ++k) {
			const int it = first_iteration + k;
			const bool err = (error_every > 0 && it % error_every == 0) || (last_iteration > 0 && it == last_iteration);
			Status s = g.iterate(err, constant_w != 0);
			if (s != ST_OK) return s;
		}
		return ST_OK;
	};
	return dispatch(e, run, run);
}

int nmfamd_engine_synchronize(nmfamd_engine* e) {
	if (!e) return NMFAMD_INVALID_ARGUMENT;
	hipStream_t s = e->elem_bytes == 4 ? e->f->stream() : e->d->stream();
	return hipStreamSynchronize(s) == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

double nmfamd_engine_frobenius(nmfamd_engine* e) { return !e ? 0.0 : (e->elem_bytes == 4 ? e->f->frobenius() : e->d->frobenius()); }
double nmfamd_engine_kl_divergence(nmfamd_engine* e) { return !e ? 0.0 : (e->elem_bytes == 4 ? e->f->kl_divergence() : e->d->kl_divergence()); }
double nmfamd_engine_rmsd(nmfamd_engine* e) { return !e ? 0.0 : (e->elem_bytes == 4 ? e->f->rmsd() : e->d->rmsd()); }

int nmfamd_engine_kernel_timing(nmfamd_engine* e, int enable) {
	// enable: 0 = off, k > 0 = bracket the launches of every k-th iteration
	return dispatch(e, [&](Engine<float>& g) { g.enable_kernel_timing(enable != 0, enable); return ST_OK; },
	                   [&](Engine<double>& g) { g.enable_kernel_timing(enable != 0, enable); return ST_OK; });
}

int nmfamd_engine_kernel_timing_read(nmfamd_engine* e, double* total_ms, long* launches) {
	return dispatch(e, [&](Engine<float>& g) { g.dominant_stats(total_ms, launches); return ST_OK; },
	                   [&](Engine<double>& g) { g.dominant_stats(total_ms, launches); return ST_OK; });
}

int nmfamd_engine_kernel_timing_read2(nmfamd_engine* e, double* total_ms, long* launches, double* pair_overhead_ms) {
	if (!e) return NMFAMD_INVALID_ARGUMENT;
	if (e->elem_bytes == 4) e->f->dominant_stats(total_ms, launches, pair_overhead_ms); else e->d->dominant_stats(total_ms, launches, pair_overhead_ms);
	return NMFAMD_OK;
}

int nmfamd_engine_kernel_timing_read3(nmfamd_engine* e, double* total_ms, long* launches, double* pair_overhead_ms, double* kind_ms, long* kind_launches) {
	if (!e) return NMFAMD_INVALID_ARGUMENT;
	if (e->elem_bytes == 4) e->f->dominant_stats(total_ms, launches, pair_overhead_ms, kind_ms, kind_launches);
	else e->d->dominant_stats(total_ms, launches, pair_overhead_ms, kind_ms, kind_launches);
	return NMFAMD_OK;
}

int nmfamd_engine_geometry(const nmfamd_engine* e, nmfamd_geometry* out) {
	if (!e || !out) return NMFAMD_INVALID_ARGUMENT;
	auto fill = [&](const auto& g) {
		out->m = g.m(); out->n = g.n(); out->r = g.r(); out->padded_rank = g.rp();
		out->padded_m = g.mpad(); out->padded_n = g.npad();
		out->slabs_h = g.slabs_h(); out->slabs_w = g.slabs_w(); out->exchange_count = g.exchange_count();
		out->product_kernel = g.product_kernel(); out->resident_images = g.resident_images(); out->one_pass = g.one_pass_state();
		out->kl_blocks_w = g.kl_blocks(true); out->kl_blocks_h = g.kl_blocks(false); out->gram_k_slices = g.gram_k_slices(); out->w_col_split = g.w_col_split() ? 1 : 0;
		out->sparse_setup = g.sparse_mode() ? (g.sparse_setup_on_device() ? 1 : 0) : -1;
		out->fused_launches = g.fused_launches(); out->gram_ride_slices_h = g.gram_ride_slices(false); out->gram_ride_slices_w = g.gram_ride_slices(true);
	};
	if (e->elem_bytes == 4) fill(*e->f); else fill(*e->d);
	return NMFAMD_OK;
}

int nmfamd_engine_geometry_sized(const nmfamd_engine* e, void* out, unsigned long struct_size) {
	if (!out || struct_size < 8) return NMFAMD_INVALID_ARGUMENT;
	nmfamd_geometry g;
	std::memset(&g, 0, sizeof(g));
	const int st = nmfamd_engine_geometry(e, &g);
	if (st != NMFAMD_OK) return st;
	std::memcpy(out, &g, struct_size < sizeof(g) ? (size_t)struct_size : sizeof(g));
	return NMFAMD_OK;
}

int nmfamd_engine_h_step(nmfamd_engine* e, int compute_error) {
	return dispatch(e, [&](Engine<float>& g) { return g.h_step(compute_error != 0); },
	                   [&](Engine<double>& g) { return g.h_step(compute_error != 0); });
}

int nmfamd_engine_set_sole_rank(nmfamd_engine* e, int sole) {
	return dispatch(e, [&](Engine<float>& g) { g.set_sole_rank(sole != 0); return ST_OK; },
	                   [&](Engine<double>& g) { g.set_sole_rank(sole != 0); return ST_OK; });
}

int nmfamd_engine_w_products(nmfamd_engine* e, void* exchange) {
	if (!exchange) return NMFAMD_INVALID_ARGUMENT;
	return dispatch(e, [&](Engine<float>& g) { return g.w_products((float*)exchange); },
	                   [&](Engine<double>& g) { return g.w_products((double*)exchange); });
}

int nmfamd_engine_w_finish(nmfamd_engine* e, const void* exchange, int compute_error) {
	if (!exchange) return NMFAMD_INVALID_ARGUMENT;
	return dispatch(e, [&](Engine<float>& g) { return g.w_finish((const float*)exchange, compute_error != 0); },
	                   [&](Engine<double>& g) { return g.w_finish((const double*)exchange, compute_error != 0); });
}

int nmfamd_engine_w_update_rows(nmfamd_engine* e, const void* num_rows, const void* hht, long row0, long rows, int compute_error, void* colsq) {
	return dispatch(e, [&](Engine<float>& g) { return g.w_update_rows((const float*)num_rows, (const float*)hht, row0, rows, compute_error != 0, (float*)colsq); },
	                   [&](Engine<double>& g) { return g.w_update_rows((const double*)num_rows, (const double*)hht, row0, rows, compute_error != 0, (double*)colsq); });
}
int nmfamd_engine_w_normalize_rows(nmfamd_engine* e, long row0, long rows, void* colsq) {
	return dispatch(e, [&](Engine<float>& g) { return g.w_normalize_rows(row0, rows, (float*)colsq); },
	                   [&](Engine<double>& g) { return g.w_normalize_rows(row0, rows, (double*)colsq); });
}
int nmfamd_engine_w_rows_replaced(nmfamd_engine* e) {
	return dispatch(e, [&](Engine<float>& g) { g.w_rows_replaced(); return ST_OK; }, [&](Engine<double>& g) { g.w_rows_replaced(); return ST_OK; });
}
void* nmfamd_engine_w_panel(nmfamd_engine* e) {
	if (!e) return nullptr;
	return e->elem_bytes == 4 ? (void*)e->f->w_panel() : (void*)e->d->w_panel();
}

long nmfamd_engine_error_terms(nmfamd_engine* e, int which, void* out, long capacity) {
	if (!e || !out || which < 0 || which > 2) return -1;
	auto copy = [&](auto& g) -> long {
		const auto& v = which == 0 ? g.terms_vtv_sorted() : (which == 1 ? g.terms_htwtv() : g.terms_hhtwtw());
		long cnt = std::min<long>((long)v.size(), capacity);
		if (cnt > 0) std::memcpy(out, v.data(), sizeof(v[0]) * (size_t)cnt);
		return cnt;
	};
	return e->elem_bytes == 4 ? copy(*e->f) : copy(*e->d);
}

long nmfamd_engine_error_terms_to_device(nmfamd_engine* e, void* dst_device, long capacity) {
	if (!e || !dst_device) return -1;
	return e->elem_bytes == 4 ? e->f->error_terms_to_device((float*)dst_device, capacity) : e->d->error_terms_to_device((double*)dst_device, capacity);
}

int nmfamd_host_kmeans_f32(const float* data, long ld, int m, int n, float* clusters, long ldc, int k, unsigned* membership, unsigned seed,
                           unsigned maxiter, double threshold, unsigned* iterations) {
	return host_kmeans<float>(data, ld, m, n, clusters, ldc, k, membership, seed, maxiter, threshold, iterations);
}
int nmfamd_host_kmeans_f64(const double* data, long ld, int m, int n, double* clusters, long ldc, int k, unsigned* membership, unsigned seed,
                           unsigned maxiter, double threshold, unsigned* iterations) {
	return host_kmeans<double>(data, ld, m, n, clusters, ldc, k, membership, seed, maxiter, threshold, iterations);
}
int nmfamd_host_init_f32(const float* V, long ldv, int m, int n, int r, int method, unsigned seed, float* W, float* H) {
	return host_init<float>(V, ldv, m, n, r, method, seed, W, H);
}
int nmfamd_host_init_f64(const double* V, long ldv, int m, int n, int r, int method, unsigned seed, double* W, double* H) {
	return host_init<double>(V, ldv, m, n, r, method, seed, W, H);
}

double nmfamd_resolve_frobenius_f32(const float* vtv_sorted, long n_vtv, float* htwtv, long n_htwtv, float* hhtwtw, long n_hhtwtw) {
	std::vector<float> a(vtv_sorted, vtv_sorted + n_vtv), b(htwtv, htwtv + n_htwtv), c(hhtwtw, hhtwtw + n_hhtwtw);
	double f = resolve_frobenius<float>(a, b, c);
	std::copy(b.begin(), b.end(), htwtv); std::copy(c.begin(), c.end(), hhtwtw);
	return f;
}

double nmfamd_resolve_frobenius_f64(const double* vtv_sorted, long n_vtv, double* htwtv, long n_htwtv, double* hhtwtw, long n_hhtwtw) {
	std::vector<double> a(vtv_sorted, vtv_sorted + n_vtv), b(htwtv, htwtv + n_htwtv), c(hhtwtw, hhtwtw + n_hhtwtw);
	double f = resolve_frobenius<double>(a, b, c);
	std::copy(b.begin(), b.end(), htwtv); std::copy(c.begin(), c.end(), hhtwtw);
	return f;
}

int nmfamd_engine_debug_read(nmfamd_engine* e, int which, void* out, long count) {
	return dispatch(e, [&](Engine<float>& g) { return g.debug_read(which, (float*)out, count); },
	                   [&](Engine<double>& g) { return g.debug_read(which, (double*)out, count); });
}

// ---- single operations on host data -------------------------------------------------------


// Tuning / diagnosis of the dominant kernel on synthetic device data: `reps` back-to-back launches
// timed with one event pair; optionally one more launch of the stamped build.
int nmfamd_tune_factor_product(int X, int Y, int reps, double* avg_us, unsigned long long* stamps_out, long stamps_capacity, long* stamps_count) {
	if (X <= 0 || Y <= 0 || reps <= 0 || !avg_us) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = 64;
	int dev = 0; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMFAMD_HIP_ERROR;
	FactorProductPlan plan = plan_factor_product(X, Y, RP, prop.multiProcessorCount);
	const long Xp = pad128(std::max<long>(X, (long)plan.xtiles * plan.th)), Yp = pad128(Y);
	DevBuf dA, dF, dS, dT;
	const long slab_stride = (long)RP * Xp;
	const long nwaves = (long)plan.xtiles * plan.splits * 8;
	if (dA.alloc(sizeof(float) * Xp * Yp) != hipSuccess || dF.alloc(sizeof(float) * RP * Yp) != hipSuccess ||
	    dS.alloc(sizeof(float) * slab_stride * plan.splits) != hipSuccess || dT.alloc(sizeof(unsigned long long) * 8 * nwaves) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (launch_fill_uniform<float>((float*)dA.p, (int)Xp, (int)Xp, Yp, Yp, 1, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;   // Xp x Yp uniform (0,1]
	if (launch_fill_uniform<float>((float*)dF.p, RP, RP, Yp, Yp, 2, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return NMFAMD_HIP_ERROR;
	for (int i = 0; i < 3; ++i) launch_factor_product_f32(plan, (const float*)dA.p, plan.th * Yp, (const float*)dF.p, RP, (float*)dS.p, slab_stride, nullptr);
	(void)hipEventRecord(e0, nullptr);
	for (int i = 0; i < reps; ++i) launch_factor_product_f32(plan, (const float*)dA.p, plan.th * Yp, (const float*)dF.p, RP, (float*)dS.p, slab_stride, nullptr);
	(void)hipEventRecord(e1, nullptr);
	if (hipEventSynchronize(e1) != hipSuccess) return NMFAMD_HIP_ERROR;
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	*avg_us = ms * 1e3 / reps;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	if (stamps_out && stamps_capacity >= 8 * nwaves) {
		if (launch_factor_product_f32_stamped(plan, (const float*)dA.p, plan.th * Yp, (const float*)dF.p, RP, (float*)dS.p, slab_stride, (unsigned long long*)dT.p, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
		if (hipMemcpy(stamps_out, dT.p, sizeof(unsigned long long) * 8 * nwaves, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
		if (stamps_count) *stamps_count = nwaves;
	} else if (stamps_count) *stamps_count = 0;
	return NMFAMD_OK;
}

int nmfamd_op_factor_product_f32(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo, int use_valu, int* out_slabs) {
	return op_factor_product<float>(A, lda, X, Y, F, ldf, r, OUT, ldo, use_valu != 0, out_slabs);
}

int nmfamd_op_factor_product_bf16(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo) {
	if (!A || !F || !OUT || X <= 0 || Y <= 0 || r <= 0 || lda < X || ldf < r || ldo < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	int dev = 0; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMFAMD_HIP_ERROR;
	const int RP = padded_rank(r);
	const long Xp = pad128(X), Yp = pad128(Y);
	const int KS = (Y + 15) / 16;
	FactorProductPlan plan; plan.th = 128; plan.xtiles = (int)(Xp / 128); plan.steps_total = KS; plan.nb = 2; plan.chunks = 1;
	plan.splits = plan_splits_bf16(plan.xtiles, KS, RP, prop.multiProcessorCount);
	DevBuf dA, dF, dAb, dFb, dS, dO;
	const long slab_stride = (long)RP * Xp;
	if (dA.alloc(sizeof(float) * Xp * Yp) != hipSuccess || dF.alloc(sizeof(float) * RP * Yp) != hipSuccess ||
	    dAb.alloc(16 * (size_t)plan.xtiles * KS * 256) != hipSuccess || dFb.alloc(16 * (size_t)KS * (RP / 32) * 64) != hipSuccess ||
	    dS.alloc(sizeof(float) * slab_stride * plan.splits) != hipSuccess || dO.alloc(sizeof(float) * slab_stride) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemcpy2D(dA.p, Xp * sizeof(float), A, lda * sizeof(float), X * sizeof(float), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dF.p, RP * sizeof(float), F, ldf * sizeof(float), r * sizeof(float), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_pack_stream_bf16((const float*)dA.p, Xp, X, Y, false, dAb.p, plan.xtiles, KS, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_pack_panel_bf16((const float*)dF.p, RP, Y, dFb.p, KS, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_factor_product_bf16(plan, dAb.p, KS, dFb.p, RP, (float*)dS.p, slab_stride, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_reduce_slabs<float>((const float*)dS.p, plan.splits, slab_stride, (float*)dO.p, slab_stride, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(OUT, ldo * sizeof(float), dO.p, RP * sizeof(float), r * sizeof(float), X, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

// x16: the x-tiled image in 16-row tiles (the engine's ONE resident image, read along its output index)
static int op_factor_product_x3(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo, int reps, double* avg_us, bool y_tiled, bool x16 = false) {
	if (!A || !F || !OUT || X <= 0 || Y <= 0 || r <= 0 || lda < X || ldf < r || ldo < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	int dev = 0; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMFAMD_HIP_ERROR;
	const int RP = padded_rank(r);
	const long Xp = pad128(X), Yp = pad128(Y);
	const int KS = (Y + 15) / 16;
	FactorProductPlan plan; plan.th = 128; plan.xtiles = (int)(Xp / 128); plan.steps_total = KS; plan.nb = 2; plan.chunks = 1;
	plan.splits = plan_splits_x3(plan.xtiles, KS, prop.multiProcessorCount);
	DevBuf dA, dT, dF, dFb, dS, dO;
	const long slab_stride = (long)RP * Xp;
	if (dA.alloc(sizeof(float) * Xp * Yp) != hipSuccess || dT.alloc(sizeof(float) * Xp * Yp) != hipSuccess || dF.alloc(sizeof(float) * RP * Yp) != hipSuccess ||
	    dFb.alloc(3 * 16 * (size_t)(KS + 1) * (RP / 32) * 64) != hipSuccess ||
	    dS.alloc(sizeof(float) * slab_stride * plan.splits) != hipSuccess || dO.alloc(sizeof(float) * slab_stride) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemset(dA.p, 0, sizeof(float) * Xp * Yp) != hipSuccess || hipMemset(dT.p, 0, sizeof(float) * Xp * Yp) != hipSuccess ||
	    hipMemset(dF.p, 0, sizeof(float) * RP * Yp) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dA.p, Xp * sizeof(float), A, lda * sizeof(float), X * sizeof(float), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dF.p, RP * sizeof(float), F, ldf * sizeof(float), r * sizeof(float), Y, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	// x-tiled image of A (128-row tiles), or (y_tiled) the image tiled along the reduction index in 16-row tiles = the x-tiled image of A^T with tile height 16:
	// the form W^T V runs in on the engine's ONE resident image of V
	const int ith = (y_tiled || x16) ? 16 : 128;
	const long tstride = y_tiled ? 16 * Xp : (x16 ? 16 * Yp : 128 * Yp);
	if (y_tiled) { if (launch_tile_transposed<float>((const float*)dA.p, Xp, X, Y, (float*)dT.p, tstride, 16, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR; }
	else if (launch_tile<float>((const float*)dA.p, Xp, X, Y, (float*)dT.p, tstride, ith, false, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_pack_panel_x3((const float*)dF.p, RP, Y, dFb.p, KS, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_factor_product_x3(plan, (const float*)dT.p, tstride, dFb.p, RP, (float*)dS.p, slab_stride, nullptr, nullptr, nullptr, y_tiled, ith) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_reduce_slabs<float>((const float*)dS.p, plan.splits, slab_stride, (float*)dO.p, slab_stride, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(OUT, ldo * sizeof(float), dO.p, RP * sizeof(float), r * sizeof(float), X, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (reps > 0 && avg_us) {
		// alternate between two images of A so that no launch finds its operand in the 256 MB memory-side cache
		// (inside an iteration the two products stream V and V^T in turn)
		DevBuf dT2;
		if (dT2.alloc(sizeof(float) * Xp * Yp) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
		if (hipMemcpy(dT2.p, dT.p, sizeof(float) * Xp * Yp, hipMemcpyDeviceToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
		hipEvent_t e0, e1;
		if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return NMFAMD_HIP_ERROR;
		for (int i = 0; i < 4; ++i) (void)launch_factor_product_x3(plan, (const float*)((i & 1) ? dT2.p : dT.p), tstride, dFb.p, RP, (float*)dS.p, slab_stride, nullptr, nullptr, nullptr, y_tiled, ith);
		(void)hipEventRecord(e0, nullptr);
		for (int i = 0; i < reps; ++i) (void)launch_factor_product_x3(plan, (const float*)((i & 1) ? dT2.p : dT.p), tstride, dFb.p, RP, (float*)dS.p, slab_stride, nullptr, nullptr, nullptr, y_tiled, ith);
		(void)hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess) return NMFAMD_HIP_ERROR;
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
		*avg_us = ms * 1e3 / reps;
		(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	}
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

int nmfamd_op_factor_product_x3(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo, int reps, double* avg_us) {
	return op_factor_product_x3(A, lda, X, Y, F, ldf, r, OUT, ldo, reps, avg_us, false);
}

int nmfamd_op_factor_product_x3_ytiled(const float* A, long lda, int X, int Y, const float* F, long ldf, int r, float* OUT, long ldo, int reps, double* avg_us) {
	return op_factor_product_x3(A, lda, X, Y, F, ldf, r, OUT, ldo, reps, avg_us, true);
}


int nmfamd_tune_factor_product_x3(int X, int Y, unsigned long long* stamps_out, long stamps_capacity, long* waves) {
	// X x Y = rows x columns of V (config 2: 10 000 x 5 000).  ONE image in 16-row tiles, as the engine keeps it; the two production forms alternate on it as they do
	// in an iteration (W^T V y-tiled, V H^T x-tiled), so the stamped launch finds the memory-side cache in the state the iteration leaves it in.
	// NMFAMD_X3_VARIANT = 10..13: the x-tiled launch is stamped; 30..33: the y-tiled one (kernels_x3.hip).
	if (X <= 0 || Y <= 0 || !stamps_out || !waves) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	int dev = 0; hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMFAMD_HIP_ERROR;
	const int RP = 64;
	const long Mp = pad128(X), Np = pad128(Y);
	const int ksW = (Y + 15) / 16, ksH = (X + 15) / 16;
	FactorProductPlan planW, planH;                  // V H^T: x = rows of V; W^T V: x = columns of V
	planW.th = planH.th = 128; planW.nb = planH.nb = 2; planW.chunks = planH.chunks = 1;
	planW.xtiles = (int)(Mp / 128); planW.steps_total = ksW; planW.splits = plan_splits_x3(planW.xtiles, ksW, prop.multiProcessorCount);
	planH.xtiles = (int)(Np / 128); planH.steps_total = ksH; planH.splits = plan_splits_x3(planH.xtiles, ksH, prop.multiProcessorCount);
	const char* ve = tuning_env("NMFAMD_X3_VARIANT");
	const bool ytiled = ve != nullptr && std::atoi(ve) >= 30;
	const FactorProductPlan& sp = ytiled ? planH : planW;
	const long nwaves = (long)sp.xtiles * sp.splits * 4;
	if (stamps_capacity < 8 * nwaves) return NMFAMD_INVALID_ARGUMENT;
	DevBuf dA, dW, dH, dWx, dHx, dS, dT;
	const long slab_stride = (long)RP * std::max(Mp, Np);
	if (dA.alloc(sizeof(float) * Mp * Np) != hipSuccess || dW.alloc(sizeof(float) * RP * Mp) != hipSuccess || dH.alloc(sizeof(float) * RP * Np) != hipSuccess ||
	    dWx.alloc(3 * 16 * (size_t)(ksH + 1) * (RP / 32) * 64) != hipSuccess || dHx.alloc(3 * 16 * (size_t)(ksW + 1) * (RP / 32) * 64) != hipSuccess ||
	    dS.alloc(sizeof(float) * slab_stride * std::max(planW.splits, planH.splits)) != hipSuccess ||
	    dT.alloc(sizeof(unsigned long long) * 8 * nwaves) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	// (any buffer of the right size is an image of random data)
	if (launch_fill_uniform<float>((float*)dA.p, (int)Mp, (int)Mp, Np, Np, 1, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_fill_uniform<float>((float*)dW.p, RP, RP, Mp, Mp, 2, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_fill_uniform<float>((float*)dH.p, RP, RP, Np, Np, 3, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_pack_panel_x3((const float*)dW.p, RP, X, dWx.p, ksH, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_pack_panel_x3((const float*)dH.p, RP, Y, dHx.p, ksW, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemset(dT.p, 0, sizeof(unsigned long long) * 8 * nwaves) != hipSuccess) return NMFAMD_HIP_ERROR;
	const long stride = 16 * Np;
	for (int i = 0; i < 40; ++i) {
		const bool last = i == 39;
		unsigned long long* sy = last && ytiled ? (unsigned long long*)dT.p : nullptr;
		unsigned long long* sx = last && !ytiled ? (unsigned long long*)dT.p : nullptr;
		if (launch_factor_product_x3(planH, (const float*)dA.p, stride, dWx.p, RP, (float*)dS.p, slab_stride, nullptr, nullptr, sy, true, 16) != hipSuccess) return NMFAMD_HIP_ERROR;
		if (launch_factor_product_x3(planW, (const float*)dA.p, stride, dHx.p, RP, (float*)dS.p, slab_stride, nullptr, nullptr, sx, false, 16) != hipSuccess) return NMFAMD_HIP_ERROR;
	}
	if (hipMemcpy(stamps_out, dT.p, sizeof(unsigned long long) * 8 * nwaves, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	*waves = nwaves;
	return NMFAMD_OK;
}

int nmfamd_op_factor_product_f64(const double* A, long lda, int X, int Y, const double* F, long ldf, int r, double* OUT, long ldo, int use_valu, int* out_slabs) {
	return op_factor_product<double>(A, lda, X, Y, F, ldf, r, OUT, ldo, use_valu != 0, out_slabs);
}

int nmfamd_op_gram_f32(const float* P, long ldp, int r, int len, float* G, long ldg) {
	if (!P || !G || r <= 0 || len <= 0 || ldp < r || ldg < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r), parts = 128;
	const long lp = pad128(len);
	DevBuf dP, dPart, dG;
	if (dP.alloc(sizeof(float) * RP * lp) != hipSuccess || dPart.alloc(sizeof(float) * (size_t)RP * RP * parts) != hipSuccess || dG.alloc(sizeof(float) * RP * RP) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemset(dP.p, 0, sizeof(float) * RP * lp) != hipSuccess) return NMFAMD_HIP_ERROR;   // the panel invariant: padding rows and columns are zero
	if (hipMemcpy2D(dP.p, RP * sizeof(float), P, ldp * sizeof(float), r * sizeof(float), len, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_gram<float>((const float*)dP.p, RP, len, parts, (float*)dPart.p, (float*)dG.p, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(G, ldg * sizeof(float), dG.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

int nmfamd_op_gram_f64(const double* P, long ldp, int r, int len, double* G, long ldg) {
	if (!P || !G || r <= 0 || len <= 0 || ldp < r || ldg < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r, sizeof(double)), parts = 128;
	const long lp = pad128(len);
	DevBuf dP, dPart, dG;
	if (dP.alloc(sizeof(double) * RP * lp) != hipSuccess || dPart.alloc(sizeof(double) * (size_t)RP * RP * parts) != hipSuccess || dG.alloc(sizeof(double) * RP * RP) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemcpy2D(dP.p, RP * sizeof(double), P, ldp * sizeof(double), r * sizeof(double), len, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_gram<double>((const double*)dP.p, RP, len, parts, (double*)dPart.p, (double*)dG.p, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(G, ldg * sizeof(double), dG.p, RP * sizeof(double), r * sizeof(double), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

int nmfamd_op_inverse_f32(const float* A, long lda, int r, float offdiag, float diag, float* Ainv, long ldi) {
	if (!A || !Ainv || r <= 0 || lda < r || ldi < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r);
	DevBuf dA, dI, dW;
	if (dA.alloc(sizeof(float) * RP * RP) != hipSuccess || dI.alloc(sizeof(float) * RP * RP) != hipSuccess || dW.alloc(sizeof(double) * 2 * (size_t)r * r) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemcpy2D(dA.p, RP * sizeof(float), A, lda * sizeof(float), r * sizeof(float), r, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_inverse_small<float>((float*)dA.p, RP, r, (float*)dI.p, (double*)dW.p, offdiag, diag, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(Ainv, ldi * sizeof(float), dI.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

// Test / measurement entry of kernels_tri.hip (padded rank 256): the passes between a factor update and the next product.
// P: [len][ldp] panel rows; colsq (r values) or NULL; theta: nsNMF smoothing.  Outputs (any may be NULL): P_out = the panel after the optional
// column normalisation; pack_out [len][r] = the bf16 fragments of the smoothed panel, widened back to fp32; G_raw / G_smooth [r][r].
// One multiplicative update of a panel at padded rank 256 in the form the bf16 path uses it (PanelTriExtras): old values with a pending column scale, optional
// scale + smoothing of the numerator rows, new rows written unnormalised together with their bf16 fragments; then the pending scale of the new panel and the
// Gram matrix of its rounded rows.  All matrices row-major [len][r] / [r][r].
int nmfamd_op_tri_update_f32(const float* P, const float* num, const float* Q, int r, int len, const float* old_colsq, int transform_num, const float* num_colsq,
                             float theta, float frag_theta, int transform_den, float* P_out, float* pack_out, float* scale_out, float* gram_out,
                             float* gram_raw_out, float* gram_image_out, float* diag_out) {
	if (!P || !num || !Q || r <= 0 || len <= 0) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r);
	if (!tri_kernels_available(RP)) return NMFAMD_INVALID_ARGUMENT;
	const long lp = pad128(len), KS = (len + 15) / 16;
	const size_t pack_bytes = 16 * (size_t)KS * (RP / 32) * 64;
	int dev = 0, cus = 256;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
	const int parts = panel_update_parts(RP, sizeof(float), (int)lp);
	DevBuf dP, dN, dQ, dOs, dNs, dPack, dPart, dG, dSq, dStage, dScale, dQx, dDummy, dG2, dImg, dDiag;
	const size_t img_bytes = (size_t)3 * 16 * (RP / 16 + 1) * (RP / 32) * 64;
	if (dG2.alloc(sizeof(float) * RP * RP) != hipSuccess || dImg.alloc(img_bytes) != hipSuccess || dDiag.alloc(sizeof(float) * RP) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (dP.alloc(sizeof(float) * RP * lp) != hipSuccess || dN.alloc(sizeof(float) * RP * lp) != hipSuccess || dQ.alloc(sizeof(float) * RP * RP) != hipSuccess ||
	    dOs.alloc(sizeof(float) * RP) != hipSuccess || dNs.alloc(sizeof(float) * RP) != hipSuccess || dPack.alloc(pack_bytes) != hipSuccess ||
	    dPart.alloc(sizeof(float) * (size_t)gram_tri_partial_elems(cus)) != hipSuccess || dG.alloc(sizeof(float) * RP * RP) != hipSuccess ||
	    dSq.alloc(sizeof(float) * ((size_t)parts + 16) * RP) != hipSuccess || dStage.alloc(sizeof(float) * RP * colsq_stage_parts()) != hipSuccess ||
	    dScale.alloc(sizeof(float) * RP) != hipSuccess || dDummy.alloc(sizeof(float) * RP * 4) != hipSuccess || dQx.alloc((size_t)3 * 16 * (RP / 16 + 1) * (RP / 32) * 64) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemcpy2D(dP.p, RP * sizeof(float), P, r * sizeof(float), r * sizeof(float), len, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dN.p, RP * sizeof(float), num, r * sizeof(float), r * sizeof(float), len, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dQ.p, RP * sizeof(float), Q, r * sizeof(float), r * sizeof(float), r, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (old_colsq && hipMemcpy(dOs.p, old_colsq, sizeof(float) * r, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (num_colsq && hipMemcpy(dNs.p, num_colsq, sizeof(float) * r, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	PanelTriExtras ex;
	ex.num_transform = transform_num != 0;
	ex.num_colsq = num_colsq ? (const float*)dNs.p : nullptr; ex.num_colsq_parts = 1;
	const float off = theta / (float)(unsigned)r, diag = (float)((1.0 - theta) + off);
	ex.num_a = diag - off; ex.num_b = off; ex.r = r;
	ex.old_colsq = old_colsq ? (const float*)dOs.p : nullptr; ex.old_colsq_parts = 1;
	ex.frag_out = dPack.p; ex.frag_KS = KS;
	const float foff = frag_theta / (float)(unsigned)r, fdiag = (float)((1.0 - frag_theta) + foff);
	if (frag_theta != 0.0f) { ex.frag_a = fdiag - foff; ex.frag_b = foff; }
	if (transform_den) { ex.den_transform = true; ex.den_a = ex.num_a; ex.den_b = ex.num_b; ex.den_colsq = ex.num_colsq; ex.den_colsq_parts = 1; }
	if (launch_panel_update<float>(PANEL_MU, (float*)dP.p, (const float*)dN.p, 1, 0, (const float*)dQ.p, RP, (int)lp, std::numeric_limits<float>::epsilon(), nullptr, len,
	                               (float*)dSq.p, nullptr, nullptr, nullptr, nullptr, 0, dQx.p, &ex) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_colsq_stage((const float*)dSq.p, RP, parts, (float*)dStage.p, nullptr, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_scale_panel_tri((float*)dDummy.p, RP, 4, (const float*)dStage.p, colsq_stage_parts(), (float*)dScale.p, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (launch_gram_tri_bf16(dPack.p, RP, KS, cus, (float*)dPart.p, (float*)dG.p, (const float*)dStage.p, colsq_stage_parts(), cus, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	// the reduction that also leaves the split image and the diagonal (what the engine's H step consumes)
	if (launch_gram_tri_bf16_image(dPack.p, RP, KS, cus, (float*)dPart.p, (float*)dG2.p, dImg.p, (float*)dDiag.p, cus, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (gram_raw_out && hipMemcpy2D(gram_raw_out, r * sizeof(float), dG2.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (diag_out && hipMemcpy(diag_out, dDiag.p, sizeof(float) * r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (gram_image_out) {
		// image element: plane p of A(c, k) at bf16 index ((((ks * NBT + nb) * 3 + p) * 2 + h) * 32 + (c & 31)) * 8 + (k & 7), ks = k / 16, nb = c / 32, h = (k / 8) & 1; A(c, k) = G(k, c)
		std::vector<uint16_t> h(img_bytes / 2);
		if (hipMemcpy(h.data(), dImg.p, img_bytes, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
		const int NBT = RP / 32;
		for (int k = 0; k < r; ++k)
			for (int c = 0; c < r; ++c) {
				float sum = 0.f;
				for (int pl = 2; pl >= 0; --pl) {
					const size_t idx = (((((size_t)(k / 16) * NBT + c / 32) * 3 + pl) * 2 + ((k / 8) & 1)) * 32 + (c & 31)) * 8 + (k & 7);
					const uint32_t bits = (uint32_t)h[idx] << 16;
					float f;
					std::memcpy(&f, &bits, 4);
					sum += f;
				}
				gram_image_out[(size_t)k * r + c] = sum;
			}
	}
	if (P_out && hipMemcpy2D(P_out, r * sizeof(float), dP.p, RP * sizeof(float), r * sizeof(float), len, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (scale_out && hipMemcpy(scale_out, dScale.p, sizeof(float) * r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (gram_out && hipMemcpy2D(gram_out, r * sizeof(float), dG.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (pack_out) {
		std::vector<uint16_t> h(pack_bytes / 2);
		if (hipMemcpy(h.data(), dPack.p, pack_bytes, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
		const int NBT = RP / 32;
		for (long y = 0; y < len; ++y)
			for (int c = 0; c < r; ++c) {
				const long frag = ((y / 16) * NBT + c / 32) * 64 + ((y / 8) & 1) * 32 + (c & 31);
				const uint32_t bits = (uint32_t)h[frag * 8 + (y & 7)] << 16;
				float f;
				std::memcpy(&f, &bits, 4);
				pack_out[y * r + c] = f;
			}
	}
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

int nmfamd_op_factor_passes_f32(const float* P, long ldp, int r, int len, const float* colsq, float theta, float* P_out, float* pack_out,
                                float* G_raw, float* G_smooth, int reps, double* avg_us_finish, double* avg_us_gram) {
	if (!P || r <= 0 || len <= 0 || ldp < r) return NMFAMD_INVALID_ARGUMENT;
	if (nmfamd_device_count() <= 0) return NMFAMD_NO_DEVICE;
	const int RP = padded_rank(r);
	if (!tri_kernels_available(RP)) return NMFAMD_INVALID_ARGUMENT;
	const long lp = pad128(len), KS = (len + 15) / 16;
	const size_t pack_bytes = 16 * (size_t)KS * (RP / 32) * 64;
	int dev = 0, cus = 256;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
	DevBuf dP, dP0, dSq, dPack, dPart, dG, dGs;
	if (dP.alloc(sizeof(float) * RP * lp) != hipSuccess || dP0.alloc(sizeof(float) * RP * lp) != hipSuccess || dSq.alloc(sizeof(float) * RP) != hipSuccess ||
	    dPack.alloc(pack_bytes) != hipSuccess || dPart.alloc(sizeof(float) * (size_t)gram_tri_partial_elems(cus)) != hipSuccess ||
	    dG.alloc(sizeof(float) * RP * RP) != hipSuccess || dGs.alloc(sizeof(float) * RP * RP) != hipSuccess) return NMFAMD_NO_DEVICE_MEMORY;
	if (hipMemset(dP0.p, 0, sizeof(float) * RP * lp) != hipSuccess || hipMemset(dSq.p, 0, sizeof(float) * RP) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (hipMemcpy2D(dP0.p, RP * sizeof(float), P, ldp * sizeof(float), r * sizeof(float), len, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (colsq && hipMemcpy(dSq.p, colsq, sizeof(float) * r, hipMemcpyHostToDevice) != hipSuccess) return NMFAMD_HIP_ERROR;
	const float off = theta / (float)(unsigned)r, diag = (float)((1.0 - theta) + off);
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return NMFAMD_HIP_ERROR;
	const int n_reps = reps > 0 ? reps : 1;
	float ms_finish = 0.f, ms_gram = 0.f;
	for (int it = 0; it < n_reps; ++it) {
		// (the normalisation is in place: every repetition starts from the caller's panel)
		if (hipMemcpyAsync(dP.p, dP0.p, sizeof(float) * RP * lp, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
		float ms = 0.f;
		(void)hipEventRecord(e0, nullptr);
		if (launch_finish_panel_bf16((float*)dP.p, RP, r, 0, lp, colsq ? (const float*)dSq.p : nullptr, 1, off, diag, dPack.p, KS, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
		(void)hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return NMFAMD_HIP_ERROR;
		ms_finish += ms;
		(void)hipEventRecord(e0, nullptr);
		if (launch_gram_tri((const float*)dP.p, RP, len, cus, (float*)dPart.p, (float*)dG.p, cus, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
		if (launch_smooth_gram((const float*)dG.p, (float*)dGs.p, RP, r, off, diag, nullptr, nullptr) != hipSuccess) return NMFAMD_HIP_ERROR;
		(void)hipEventRecord(e1, nullptr);
		if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return NMFAMD_HIP_ERROR;
		ms_gram += ms;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	if (avg_us_finish) *avg_us_finish = 1e3 * ms_finish / n_reps;
	if (avg_us_gram) *avg_us_gram = 1e3 * ms_gram / n_reps;
	if (P_out && hipMemcpy2D(P_out, ldp * sizeof(float), dP.p, RP * sizeof(float), r * sizeof(float), len, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (G_raw && hipMemcpy2D(G_raw, r * sizeof(float), dG.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (G_smooth && hipMemcpy2D(G_smooth, r * sizeof(float), dGs.p, RP * sizeof(float), r * sizeof(float), r, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
	if (pack_out) {
		std::vector<uint16_t> h(pack_bytes / 2);
		if (hipMemcpy(h.data(), dPack.p, pack_bytes, hipMemcpyDeviceToHost) != hipSuccess) return NMFAMD_HIP_ERROR;
		const int NBT = RP / 32;
		for (long y = 0; y < len; ++y)
			for (int c = 0; c < r; ++c) {
				const long frag = ((y / 16) * NBT + c / 32) * 64 + ((y / 8) & 1) * 32 + (c & 31);
				const uint32_t bits = (uint32_t)h[frag * 8 + (y & 7)] << 16;
				float f;
				std::memcpy(&f, &bits, 4);
				pack_out[y * r + c] = f;
			}
	}
	return hipDeviceSynchronize() == hipSuccess ? NMFAMD_OK : NMFAMD_HIP_ERROR;
}

} // extern "C"
