// comm.hip -- RCCL (C API, loaded at run time) and in-process transports of comm.h.
#include "comm.h"
#include "kernels.h"
#include "tuning.h"

#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace nmfamd {

// ---- exchange buffers: the part every transport shares (a team of one reads its own buffer) ---------------------------
Status Comm::exchange_alloc(size_t bytes, int slots, void** mine) {
	if (bytes == 0 || slots < 1 || slots > 2 || mine == nullptr) return ST_INVALID;
	// (a second sharded run on the same communicator reuses the buffers when they are large enough)
	if ((int)xbuf_.size() != slots || bytes > xbuf_bytes_) {
		for (void* p : xbuf_) if (p) (void)hipFree(p);
		xbuf_.clear(); xbuf_bytes_ = 0;
		for (int i = 0; i < slots; ++i) {
			void* p = nullptr;
			if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return ST_NO_DEVICE_MEMORY; }
			xbuf_.push_back(p);
		}
		xbuf_bytes_ = bytes;
	}
	for (int i = 0; i < slots; ++i) {
		if (hipMemset(xbuf_[i], 0, bytes) != hipSuccess) { (void)hipGetLastError(); return ST_HIP_ERROR; }
		mine[i] = xbuf_[i];
	}
	return ST_OK;
}

Status Comm::exchange_publish(int slot, hipStream_t, const void** peers) {
	if (world() != 1 || slot < 0 || slot >= (int)xbuf_.size() || peers == nullptr) return ST_INVALID;
	peers[0] = xbuf_[slot];
	return ST_OK;
}

// =====================================================================================================================
// RCCL through its C API.  The library is loaded on first use (dlopen): a single-GPU caller never pays for it, and
// libnmfgpu64.so keeps linking without RCCL installed.  ncclReduceScatter / ncclAllGather are the direct algorithms
// SURVEY.md section 5 asks for on a fully connected xGMI node (all seven links at once instead of a one-link ring).
// =====================================================================================================================
namespace {

struct RcclApi {
	void* handle = nullptr;
	std::string error;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclReduceScatter) ReduceScatter = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi& rccl_api() {
	static RcclApi api;
	static std::once_flag once;
	std::call_once(once, [] {
		const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char* n : names) {
			api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (api.handle) break;
		}
		if (!api.handle) { const char* e = dlerror(); api.error = e ? e : "librccl.so not found"; return; }
		bool ok = true;
		auto sym = [&](const char* name) -> void* {
			void* p = dlsym(api.handle, name);
			if (!p) { ok = false; api.error = std::string("missing RCCL symbol ") + name; }
			return p;
		};
		api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
		api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
		api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
		api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
		api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
		api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
		api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
		api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
		api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
		if (!ok) { dlclose(api.handle); api.handle = nullptr; }
	});
	return api;
}

class RcclComm : public Comm {
public:
	RcclComm(ncclComm_t c, int rank, int world) : comm_(c), rank_(rank), world_(world) {}
	~RcclComm() override { if (comm_) (void)rccl_api().CommDestroy(comm_); }
	int rank() const override { return rank_; }
	int world() const override { return world_; }
	const char* transport() const override { return "rccl"; }
	const char* last_error() const override { return error_.c_str(); }

	Status all_reduce(void* buf, long count, int eb, hipStream_t s) override {
		return check(rccl_api().AllReduce(buf, buf, (size_t)count, type(eb), ncclSum, comm_, s), "ncclAllReduce");
	}
	Status reduce_scatter(const void* send, void* recv, long count, int eb, hipStream_t s) override {
		return check(rccl_api().ReduceScatter(send, recv, (size_t)count, type(eb), ncclSum, comm_, s), "ncclReduceScatter");
	}
	Status all_gather_inplace(void* buf, long count, int eb, hipStream_t s) override {
		// in place: the send buffer is this rank's part of the receive buffer
		const char* mine = static_cast<const char*>(buf) + (size_t)rank_ * (size_t)count * (size_t)eb;
		return check(rccl_api().AllGather(mine, buf, (size_t)count, type(eb), comm_, s), "ncclAllGather");
	}
	void group_begin() override { (void)rccl_api().GroupStart(); }
	Status group_end() override { return check(rccl_api().GroupEnd(), "ncclGroupEnd"); }

private:
	static ncclDataType_t type(int eb) { return eb == 8 ? ncclFloat64 : ncclFloat32; }
	Status check(ncclResult_t r, const char* what) {
		if (r == ncclSuccess) return ST_OK;
		error_ = std::string(what) + ": " + rccl_api().GetErrorString(r);
		return ST_HIP_ERROR;
	}
	ncclComm_t comm_;
	int rank_, world_;
	std::string error_;
};

} // namespace

bool rccl_available(const char** why) {
	RcclApi& api = rccl_api();
	if (why) *why = api.error.c_str();
	return api.handle != nullptr;
}

Status rccl_unique_id(void* out128) {
	static_assert(sizeof(ncclUniqueId) == COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
	if (!out128 || !rccl_available()) return ST_HIP_ERROR;
	ncclUniqueId id;
	if (rccl_api().GetUniqueId(&id) != ncclSuccess) return ST_HIP_ERROR;
	std::memcpy(out128, &id, sizeof(id));
	return ST_OK;
}

Status rccl_comm_create(const void* id128, int world, int rank, std::unique_ptr<Comm>* out) {
	if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return ST_INVALID;
	if (!rccl_available()) return ST_HIP_ERROR;
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof(id));
	ncclComm_t c = nullptr;
	if (rccl_api().CommInitRank(&c, world, id, rank) != ncclSuccess) return ST_HIP_ERROR;
	out->reset(new RcclComm(c, rank, world));
	return ST_OK;
}

// =====================================================================================================================
// In-process transport: the rank threads publish their buffers, every rank's kernel reads the peers' buffers directly
// and adds them in rank order (so every rank gets the same bits).  Stream ordering between ranks goes through HIP
// events (recorded by the owner of a buffer, waited for by every reader), host ordering through a spinning barrier.
// =====================================================================================================================
constexpr int LOCAL_MAX_WORLD = 16;

struct LocalGroup {
	int world = 1;
	std::atomic<int> arrived{0};
	std::atomic<unsigned> generation{0};
	std::atomic<bool> aborted{false};
	std::mutex failure_lock;
	std::string failure;                 // why the set-up failed (the first rank to fail says which pair of devices): local_group_failure()
	std::string selftest_report;         // one line of the set-up self-test (rank 0's): local_group_selftest()
	void fail(const std::string& what) { std::lock_guard<std::mutex> g(failure_lock); if (failure.empty()) failure = what; }
	struct Slot {
		const void* buf = nullptr;
		hipEvent_t ready = nullptr, done = nullptr;
		int device = -1;
		// exchange by direct reads (Comm::exchange_alloc / exchange_publish): this rank's two buffers and the events behind their writers.
		// The GROUP owns the buffers: peers' kernels read them, so they are released only when the last rank has let go of the group -- and a
		// rank lets go (closes its communicator) only after its own device has drained (~LocalComm), i.e. after its last reader of anybody's buffer.
		const void* xbuf[2] = {nullptr, nullptr};
		void* xown[2] = {nullptr, nullptr};
		size_t xbytes = 0;
		hipEvent_t xready[2] = {nullptr, nullptr};
	};
	std::vector<Slot> slots;
	// a rendezvous that does NOT give way to the abort flag (every rank reaches it on every path): used where a rank must not free what a peer's kernel may still read
	std::atomic<int> hard_arrived{0};
	std::atomic<unsigned> hard_generation{0};
	~LocalGroup() {
		for (Slot& s : slots) {
			if (s.ready) (void)hipEventDestroy(s.ready);
			if (s.done) (void)hipEventDestroy(s.done);
			for (hipEvent_t e : s.xready) if (e) (void)hipEventDestroy(e);
			for (void* p : s.xown) if (p) (void)hipFree(p);
		}
		(void)hipGetLastError();
	}
};

std::shared_ptr<LocalGroup> local_group_create(int world) {
	if (world < 1 || world > LOCAL_MAX_WORLD) return nullptr;
	auto g = std::make_shared<LocalGroup>();
	g->world = world;
	g->slots.resize(world);
	return g;
}

// false: the group was aborted (a rank failed); every waiting rank leaves
static bool barrier(LocalGroup& g) {
	if (g.world == 1) return !g.aborted.load(std::memory_order_acquire);
	const unsigned gen = g.generation.load(std::memory_order_acquire);
	if (g.arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == g.world) {
		g.arrived.store(0, std::memory_order_relaxed);
		g.generation.store(gen + 1, std::memory_order_release);
	} else {
		unsigned spins = 0;
		while (g.generation.load(std::memory_order_acquire) == gen) {
			if (g.aborted.load(std::memory_order_acquire)) return false;
			if (++spins > 4000) std::this_thread::yield();
		}
	}
	return !g.aborted.load(std::memory_order_acquire);
}

// every rank arrives, aborted or not (bounded: a rank that never comes -- a crashed thread -- must not hang its peers for ever: ~20 s, then false)
static bool barrier_hard(LocalGroup& g) {
	if (g.world == 1) return true;
	const unsigned gen = g.hard_generation.load(std::memory_order_acquire);
	if (g.hard_arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == g.world) {
		g.hard_arrived.store(0, std::memory_order_relaxed);
		g.hard_generation.store(gen + 1, std::memory_order_release);
		return true;
	}
	const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(20);
	unsigned spins = 0;
	while (g.hard_generation.load(std::memory_order_acquire) == gen) {
		if (++spins > 4000) { std::this_thread::yield(); if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() > t_end) return false; }
	}
	return true;
}

void local_group_barrier(LocalGroup& g) { (void)barrier(g); }
std::string local_group_failure(LocalGroup& g) { std::lock_guard<std::mutex> l(g.failure_lock); return g.failure; }
void local_group_abort(LocalGroup& g) { g.aborted.store(true, std::memory_order_release); }

namespace {

struct PeerPtrs { const void* p[LOCAL_MAX_WORLD]; };

// out[i] = sum over ranks (ascending) of peer_p[offset + i]; 16 bytes per lane and load.
// A BOUNDED grid (LOCAL_COPY_BLOCKS workgroups, grid-stride, four independent vectors per lane and turn): the kernels open with a system-scope acquire -- buffers in other
// devices' memory are re-read at the same addresses every iteration, and whatever fence scope the runtime gave the launch, lines cached two iterations ago must not be served
// again -- and that fence is paid per WAVE: with one workgroup per 4 KiB (round 4's first form) the 51 MB reduce-scatter of a config-4 shard took 358 us instead of ~25
// (200 000 fences; tools/c4_shard_modes.py).  A team of one has no peer memory and skips it.
constexpr int LOCAL_COPY_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void k_local_sum(PeerPtrs peers, int world, long offset, T* __restrict__ out, long count) {
	constexpr int V = 16 / sizeof(T);
	typedef T vec __attribute__((ext_vector_type(V)));
	if (world > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	const long stride = (long)gridDim.x * 256 * V;
	long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;
	for (; i + 3 * stride + V <= count; i += 4 * stride) {
		vec s[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) s[u] = *reinterpret_cast<const vec*>(static_cast<const T*>(peers.p[0]) + offset + i + u * stride);
		for (int p = 1; p < world; ++p) {
			vec t[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec*>(static_cast<const T*>(peers.p[p]) + offset + i + u * stride);
#pragma unroll
			for (int u = 0; u < 4; ++u) s[u] += t[u];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) *reinterpret_cast<vec*>(out + i + u * stride) = s[u];
	}
	for (; i < count; i += stride) {
		if (i + V <= count) {
			vec s = *reinterpret_cast<const vec*>(static_cast<const T*>(peers.p[0]) + offset + i);
			for (int p = 1; p < world; ++p) s += *reinterpret_cast<const vec*>(static_cast<const T*>(peers.p[p]) + offset + i);
			*reinterpret_cast<vec*>(out + i) = s;
		} else {
			for (long k = i; k < count; ++k) {
				T s = static_cast<const T*>(peers.p[0])[offset + k];
				for (int p = 1; p < world; ++p) s += static_cast<const T*>(peers.p[p])[offset + k];
				out[k] = s;
			}
		}
	}
}

// mine[p * count + i] = peer_p[p * count + i] for every peer p != rank (grid.y = peer; grid.x bounded as above)
template <typename T>
__global__ __launch_bounds__(256) void k_local_gather(PeerPtrs peers, int world, int rank, T* __restrict__ mine, long count) {
	constexpr int V = 16 / sizeof(T);
	typedef T vec __attribute__((ext_vector_type(V)));
	const int p = blockIdx.y;
	if (p == rank) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (as in k_local_sum; a team of one never launches this kernel)
	const T* src = static_cast<const T*>(peers.p[p]) + (long)p * count;
	T* dst = mine + (long)p * count;
	const long stride = (long)gridDim.x * 256 * V;
	long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;
	for (; i + 3 * stride + V <= count; i += 4 * stride) {
		vec t[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec*>(src + i + u * stride);
#pragma unroll
		for (int u = 0; u < 4; ++u) *reinterpret_cast<vec*>(dst + i + u * stride) = t[u];
	}
	for (; i < count; i += stride) {
		if (i + V <= count) *reinterpret_cast<vec*>(dst + i) = *reinterpret_cast<const vec*>(src + i);
		else for (long k = i; k < count; ++k) dst[k] = src[k];
	}
}

// ---- set-up self-test of the transport (LocalComm::self_test) -----------------------------------------------------------------------------------------
// value i of the buffer rank `rank` publishes in slot `slot` during phase `phase` of the test: small integers, so that every sum over <= 16 ranks is exact in
// fp32 whatever its order, and different for every (rank, slot, phase) so that a line served from an earlier phase or from the other slot shows
__host__ __device__ inline float selftest_value(int rank, int slot, int phase, long i) { return (float)((i * 7 + rank * 13 + slot * 5 + phase * 101 + 1) % 251); }
__global__ __launch_bounds__(256) void k_selftest_fill(float* __restrict__ buf, long count, int rank, int slot, int phase) {
	const long i = (long)blockIdx.x * 256 + threadIdx.x;
	if (i < count) buf[i] = selftest_value(rank, slot, phase, i);
}
constexpr int SELFTEST_LEN = 64;                                 // panel columns of the test's exchange panel (two workgroups of the update kernel)
constexpr long SELFTEST_PANEL = 64l * SELFTEST_LEN, SELFTEST_COUNT = SELFTEST_PANEL + 4096;       // [panel 64 x LEN | r x r part]

class LocalComm : public Comm {
public:
	LocalComm(std::shared_ptr<LocalGroup> g, int rank) : g_(std::move(g)), rank_(rank) {}
	// closing: this rank's device drains first -- its kernels may be reading the peers' exchange buffers, which live until the LAST rank has let go of the group
	// (the rank's device is the one that was current when it joined the group, whatever thread closes the communicator)
	~LocalComm() override {
		int cur = -1;
		const int dev = g_->slots[rank_].device;
		if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev) (void)hipSetDevice(dev); else cur = -1;
		(void)hipDeviceSynchronize();
		if (scratch_) (void)hipFree(scratch_);
		if (cur >= 0) (void)hipSetDevice(cur);
		(void)hipGetLastError();
	}
	// First contact of the ranks' devices, BEFORE any iteration depends on it: every rank publishes pattern A, every rank reads every
	// rank's buffers THROUGH THE KERNELS THE ITERATION USES -- the W update's prologue (k_mu64_update32 with PeerSlabs) and k_sum_peers behind exchange_publish's
	// one rendezvous, two alternating slots; k_local_sum / k_local_gather behind publish / retire -- and checks every word; then pattern B at the SAME addresses
	// (what a one-device box can never show: a reader's cache serving the lines of two iterations ago).  Any mismatch names the (reader, owner) pair in the
	// group's failure text and fails the set-up on EVERY rank (the caller -- runner.h -- then forms an RCCL clique instead, or fails compute).
	// fault (measurement builds, NMFAMD_SELFTEST_FAULT = rank): that rank skips its second write.
	Status self_test(std::string* report);
	int rank() const override { return rank_; }
	int world() const override { return g_->world; }
	const char* transport() const override { return "in-process (peer reads)"; }
	const char* last_error() const override { return error_; }

	Status all_reduce(void* buf, long count, int eb, hipStream_t s) override {
		if (count <= 0) return ST_OK;
		if (g_->world == 1) return ST_OK;
		if (!aligned(buf) || !grow_scratch((size_t)count * eb)) return fail("all_reduce: buffer");
		PeerPtrs pp;
		if (Status st = publish(buf, s, pp)) return st;
		if (eb == 8) hipLaunchKernelGGL((k_local_sum<double>), blocks(count, 8), dim3(256), 0, s, pp, g_->world, 0l, static_cast<double*>(scratch_), count);
		else hipLaunchKernelGGL((k_local_sum<float>), blocks(count, 4), dim3(256), 0, s, pp, g_->world, 0l, static_cast<float*>(scratch_), count);
		if (hipGetLastError() != hipSuccess) return fail("all_reduce: launch");
		if (Status st = retire(s)) return st;          // every peer has read this rank's buffer: it may be overwritten now
		if (hipMemcpyAsync(buf, scratch_, (size_t)count * eb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("all_reduce: copy");
		return ST_OK;
	}

	Status reduce_scatter(const void* send, void* recv, long count, int eb, hipStream_t s) override {
		if (count <= 0) return ST_OK;
		if (!aligned(send) || !aligned(recv) || ((size_t)count * eb) % 16 != 0) return fail("reduce_scatter: alignment");
		PeerPtrs pp;
		if (Status st = publish(send, s, pp)) return st;
		const long off = (long)rank_ * count;
		if (eb == 8) hipLaunchKernelGGL((k_local_sum<double>), blocks(count, 8), dim3(256), 0, s, pp, g_->world, off, static_cast<double*>(recv), count);
		else hipLaunchKernelGGL((k_local_sum<float>), blocks(count, 4), dim3(256), 0, s, pp, g_->world, off, static_cast<float*>(recv), count);
		if (hipGetLastError() != hipSuccess) return fail("reduce_scatter: launch");
		return retire(s);
	}

	Status all_gather_inplace(void* buf, long count, int eb, hipStream_t s) override {
		if (count <= 0 || g_->world == 1) return ST_OK;
		if (!aligned(buf) || ((size_t)count * eb) % 16 != 0) return fail("all_gather: alignment");
		PeerPtrs pp;
		if (Status st = publish(buf, s, pp)) return st;
		if (eb == 8) hipLaunchKernelGGL((k_local_gather<double>), blocks2(count, 8), dim3(256), 0, s, pp, g_->world, rank_, static_cast<double*>(buf), count);
		else hipLaunchKernelGGL((k_local_gather<float>), blocks2(count, 4), dim3(256), 0, s, pp, g_->world, rank_, static_cast<float*>(buf), count);
		if (hipGetLastError() != hipSuccess) return fail("all_gather: launch");
		return retire(s);
	}

	bool direct_exchange() const override { return true; }
	Status exchange_alloc(size_t bytes, int slots, void** mine) override {
		if (bytes == 0 || slots != 2 || mine == nullptr) return ST_INVALID;
		LocalGroup::Slot& me = g_->slots[rank_];
		// A collective.  The buffers may be in use: a peer's last W update of a previous run on this communicator is enqueued on the PEER's stream and may still be
		// reading this rank's slot.  So before anything is cleared, released or replaced: every rank arrives, every rank's device drains, every rank
		// arrives again -- then no reader of any old buffer is left anywhere.
		if (g_->world > 1) {
			const bool a = barrier_hard(*g_);
			(void)hipDeviceSynchronize();
			const bool b = barrier_hard(*g_);
			if (!a || !b) return fail("exchange_alloc: a rank did not arrive");
		}
		Status st = ST_OK;
		if (bytes > me.xbytes) {
			for (void*& p : me.xown) { if (p) (void)hipFree(p); p = nullptr; }
			me.xbytes = 0;
			for (int i = 0; i < 2 && st == ST_OK; ++i) if (hipMalloc(&me.xown[i], bytes) != hipSuccess) { (void)hipGetLastError(); me.xown[i] = nullptr; st = ST_NO_DEVICE_MEMORY; }
			if (st == ST_OK) me.xbytes = bytes;
		}
		for (int i = 0; i < 2 && st == ST_OK; ++i) {
			if (hipMemset(me.xown[i], 0, me.xbytes) != hipSuccess) { (void)hipGetLastError(); st = ST_HIP_ERROR; break; }
			mine[i] = me.xown[i];
			me.xbuf[i] = me.xown[i];
			if (me.xready[i] == nullptr && hipEventCreateWithFlags(&me.xready[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); st = ST_HIP_ERROR; }
		}
		if (st == ST_OK && hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); st = ST_HIP_ERROR; }      // (the clearing is done before any peer may read)
		if (st != ST_OK) g_->aborted.store(true, std::memory_order_release);
		if (!barrier(*g_)) return st != ST_OK ? st : ST_HIP_ERROR;          // every rank's buffers and events exist from here on
		return ST_OK;
	}
	Status exchange_publish(int slot, hipStream_t s, const void** peers) override {
		if (slot < 0 || slot > 1 || peers == nullptr) return ST_INVALID;
		LocalGroup::Slot& me = g_->slots[rank_];
		if (me.xready[slot] == nullptr) return ST_INVALID;
		if (g_->world > 1) {
			if (hipEventRecord(me.xready[slot], s) != hipSuccess) return fail("event record");
			if (!barrier(*g_)) return ST_HIP_ERROR;                     // every rank has recorded: the waits below see THIS iteration's records
		}
		for (int p = 0; p < g_->world; ++p) {
			peers[p] = g_->slots[p].xbuf[slot];
			if (p != rank_ && hipStreamWaitEvent(s, g_->slots[p].xready[slot], 0) != hipSuccess) return fail("event wait");
		}
		return ST_OK;
	}

private:
	static bool aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
	static dim3 blocks(long count, int eb) { const long per = 256l * (16 / eb); return dim3((unsigned)std::min<long>(LOCAL_COPY_BLOCKS, (count + per - 1) / per)); }
	dim3 blocks2(long count, int eb) const { const long per = 256l * (16 / eb); return dim3((unsigned)std::min<long>(std::max(64, LOCAL_COPY_BLOCKS / g_->world), (count + per - 1) / per), (unsigned)g_->world); }
	Status fail(const char* what) { error_ = what; g_->aborted.store(true, std::memory_order_release); (void)hipGetLastError(); return ST_HIP_ERROR; }

	// This rank's buffer becomes readable by the peers once the work enqueued so far has run; returns every rank's
	// buffer, with this stream ordered behind all of them
	Status publish(const void* buf, hipStream_t s, PeerPtrs& pp) {
		LocalGroup::Slot& me = g_->slots[rank_];
		me.buf = buf;
		if (hipEventRecord(me.ready, s) != hipSuccess) return fail("event record");
		if (!barrier(*g_)) return ST_HIP_ERROR;
		for (int p = 0; p < g_->world; ++p) {
			pp.p[p] = g_->slots[p].buf;
			if (p != rank_ && hipStreamWaitEvent(s, g_->slots[p].ready, 0) != hipSuccess) return fail("event wait");
		}
		return ST_OK;
	}
	// ... and the peers' buffers stay untouched by their owners until every reader is through
	Status retire(hipStream_t s) {
		if (hipEventRecord(g_->slots[rank_].done, s) != hipSuccess) return fail("event record");
		if (!barrier(*g_)) return ST_HIP_ERROR;
		for (int p = 0; p < g_->world; ++p)
			if (p != rank_ && hipStreamWaitEvent(s, g_->slots[p].done, 0) != hipSuccess) return fail("event wait");
		return ST_OK;
	}
	bool grow_scratch(size_t bytes) {
		if (bytes <= scratch_bytes_) return true;
		if (scratch_) (void)hipFree(scratch_);
		scratch_ = nullptr; scratch_bytes_ = 0;
		if (hipMalloc(&scratch_, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
		scratch_bytes_ = bytes;
		return true;
	}

	std::shared_ptr<LocalGroup> g_;
	int rank_;
	void* scratch_ = nullptr;
	size_t scratch_bytes_ = 0;
	const char* error_ = "";
};

} // namespace

Status LocalComm::self_test(std::string* report) {
	LocalGroup& g = *g_;
	const int world = g.world;
	if (world < 2) return ST_OK;
	const auto t0 = std::chrono::steady_clock::now();
	int fault_rank = -1;
	if (const char* e = tuning_env("NMFAMD_SELFTEST_FAULT")) fault_rank = std::atoi(e);
	hipStream_t s = nullptr;
	float *P = nullptr, *P2 = nullptr, *Q = nullptr, *scale = nullptr, *ps = nullptr, *x3 = nullptr, *expect = nullptr, *sums = nullptr, *coll = nullptr, *res = nullptr;
	const long x3_floats = (long)(SELFTEST_LEN / 16 + 1) * 2 * 3 * 64 * 4;          // split image of the test panel (16 bytes per fragment)
	const long res_per_round = 2 * SELFTEST_PANEL + 4096;                         // [update through the peers | update on local copies | summed r x r part]
	const long coll_count = 4096;                                                   // elements per rank of the collectives' test buffer
	const long res_total = 4 * res_per_round + 2 * (coll_count + (long)world * coll_count + coll_count);
	std::vector<float> host;
	void* mine[2] = {nullptr, nullptr};
	std::string why;
	auto ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && why.empty()) { why = std::string(what) + ": " + hipGetErrorString(e); (void)hipGetLastError(); } return e == hipSuccess; };
	bool good = ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream");
	// (one allocation for all of the test's scratch: ten hipMalloc / hipFree pairs per rank were a third of its time)
	float* pool = nullptr;
	{
		long need = 0;
		auto take = [&](long n) { const long at = need; need += (n + 63) / 64 * 64; return at; };
		const long oP = take(SELFTEST_PANEL), oP2 = take(SELFTEST_PANEL), oQ = take(4096), oS = take(64), oPs = take(4096), oX = take(x3_floats),
		           oE = take((long)world * SELFTEST_COUNT), oSu = take(4096), oC = take((long)world * coll_count), oR = take(res_total);
		good = good && ok(hipMalloc((void**)&pool, sizeof(float) * (size_t)need), "hipMalloc");
		if (good) { P = pool + oP; P2 = pool + oP2; Q = pool + oQ; scale = pool + oS; ps = pool + oPs; x3 = pool + oX; expect = pool + oE; sums = pool + oSu; coll = pool + oC; res = pool + oR; }
	}
	Status st = good ? ST_OK : ST_NO_DEVICE_MEMORY;
	// the exchange buffers of the test ARE the transport's (exchange_alloc): the first sharded run that needs no larger ones reuses these very addresses
	if (st != ST_OK) { g.fail("rank " + std::to_string(rank_) + ": self-test set-up: " + why); g.aborted.store(true, std::memory_order_release); }
	{ Status xa = exchange_alloc(sizeof(float) * SELFTEST_COUNT, 2, mine); if (st == ST_OK) st = xa; }       // (a collective: every rank calls it, whatever its state)
	if (st != ST_OK) { (void)hipGetLastError(); }
	auto fill = [&](float* buf, long count, int rank, int slot, int phase) {
		hipLaunchKernelGGL(k_selftest_fill, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, buf, count, rank, slot, phase);
	};
	if (st == ST_OK) {
		// constants of the update: old panel of ones, Q = identity, unit column scale
		std::vector<float> ones(SELFTEST_PANEL, 1.0f), ident(4096, 0.0f);
		for (int c = 0; c < 64; ++c) ident[c * 64 + c] = 1.0f;
		good = ok(hipMemcpyAsync(Q, ident.data(), sizeof(float) * 4096, hipMemcpyHostToDevice, s), "copy") && ok(hipMemcpyAsync(scale, ones.data(), sizeof(float) * 64, hipMemcpyHostToDevice, s), "copy") &&
		       ok(hipMemsetAsync(res, 0, sizeof(float) * (size_t)res_total, s), "memset") && ok(hipStreamSynchronize(s), "sync");
		if (!good) st = ST_HIP_ERROR;
	}
	const float eps = 1.1920929e-07f;
	// ---- the exchange by direct reads: rounds (phase A, slot 0) (A, 1) (B, 0) (B, 1), one publish each, as four iterations of a sharded run ----
	for (int round = 0; round < 4 && st == ST_OK; ++round) {
		const int phase = round >> 1, slot = round & 1;
		float* r0 = res + (long)round * res_per_round;
		if (!(phase == 1 && rank_ == fault_rank)) fill(static_cast<float*>(mine[slot]), SELFTEST_COUNT, rank_, slot, phase);
		for (int p = 0; p < world; ++p) fill(expect + (long)p * SELFTEST_COUNT, SELFTEST_COUNT, p, slot, phase);          // what every rank's buffer must hold now
		const void* peers[LOCAL_MAX_WORLD];
		st = exchange_publish(slot, s, peers);
		if (st != ST_OK) break;
		PeerSlabs panels = {}, hhts = {}, lpanels = {};
		panels.count = hhts.count = lpanels.count = world;
		for (int p = 0; p < world; ++p) {
			panels.p[p] = static_cast<const float*>(peers[p]); hhts.p[p] = panels.p[p] + SELFTEST_PANEL;
			lpanels.p[p] = expect + (long)p * SELFTEST_COUNT;
		}
		good = ok(launch_sum_peers(hhts, r0 + 2 * SELFTEST_PANEL, 4096, s), "k_sum_peers");
		// the W update as Engine::w_finish_peers launches it, once through the peers' memory and once on the local copies of what they must hold
		for (int local = 0; local < 2 && good; ++local) {
			float* Pn = local ? P2 : P;
			good = ok(hipMemsetAsync(Pn, 0, sizeof(float) * SELFTEST_PANEL, s), "memset");
			if (good) { hipLaunchKernelGGL(k_selftest_fill, dim3((unsigned)(SELFTEST_PANEL / 256)), dim3(256), 0, s, Pn, SELFTEST_PANEL, 0, 0, 7); }
			good = good && ok(launch_mu64_update32(1, Pn, nullptr, world, 0, Q, scale, eps, ps, SELFTEST_LEN, SELFTEST_LEN, nullptr, 0, s, x3, SELFTEST_LEN / 16, local ? &lpanels : &panels), "k_mu64_update32");
			good = good && ok(hipMemcpyAsync(r0 + (long)local * SELFTEST_PANEL, Pn, sizeof(float) * SELFTEST_PANEL, hipMemcpyDeviceToDevice, s), "copy");
		}
		if (!good) st = ST_HIP_ERROR;
	}
	// ---- the collectives (publish / retire): all-reduce, all-gather, reduce-scatter of pattern A, then of pattern B in the same buffers ----
	float* c0 = res + 4 * res_per_round;
	for (int phase = 0; phase < 2 && st == ST_OK; ++phase) {
		float* cr = c0 + (long)phase * (coll_count + (long)world * coll_count + coll_count);
		if (!(phase == 1 && rank_ == fault_rank)) fill(coll, coll_count, rank_, 2, phase);
		st = all_reduce(coll, coll_count, 4, s);
		if (st == ST_OK && !ok(hipMemcpyAsync(cr, coll, sizeof(float) * coll_count, hipMemcpyDeviceToDevice, s), "copy")) st = ST_HIP_ERROR;
		if (st != ST_OK) break;
		if (!(phase == 1 && rank_ == fault_rank)) fill(coll + (long)rank_ * coll_count, coll_count, rank_, 3, phase);
		st = all_gather_inplace(coll, coll_count, 4, s);
		if (st == ST_OK && !ok(hipMemcpyAsync(cr + coll_count, coll, sizeof(float) * (size_t)world * coll_count, hipMemcpyDeviceToDevice, s), "copy")) st = ST_HIP_ERROR;
		if (st != ST_OK) break;
		if (!(phase == 1 && rank_ == fault_rank)) fill(coll, (long)world * coll_count, rank_, 4, phase);
		st = reduce_scatter(coll, sums, coll_count, 4, s);
		if (st == ST_OK && !ok(hipMemcpyAsync(cr + coll_count + (long)world * coll_count, sums, sizeof(float) * coll_count, hipMemcpyDeviceToDevice, s), "copy")) st = ST_HIP_ERROR;
	}
	if (st == ST_OK) {
		host.resize(res_total);
		if (!ok(hipMemcpyAsync(host.data(), res, sizeof(float) * (size_t)res_total, hipMemcpyDeviceToHost, s), "copy back") || !ok(hipStreamSynchronize(s), "sync")) st = ST_HIP_ERROR;
	}
	// ---- every word ----
	std::string bad;
	if (st == ST_OK) {
		for (int round = 0; round < 4 && bad.empty(); ++round) {
			const int phase = round >> 1, slot = round & 1;
			const float* r0 = host.data() + (long)round * res_per_round;
			for (long i = 0; i < 4096 && bad.empty(); ++i) {
				float want = 0.f;
				for (int p = 0; p < world; ++p) want += selftest_value(p, slot, phase, SELFTEST_PANEL + i);
				if (r0[2 * SELFTEST_PANEL + i] != want) {
					bad = "k_sum_peers read word " + std::to_string(i) + " of the r x r parts as a sum of " + std::to_string(r0[2 * SELFTEST_PANEL + i]) + " instead of " + std::to_string(want);
					for (int p = 0; p < world; ++p)       // a single stale owner explains it?
						for (int ph = 0; ph < 2; ++ph)
							for (int sl = 0; sl < 2; ++sl)
								if ((ph != phase || sl != slot) && want - selftest_value(p, slot, phase, SELFTEST_PANEL + i) + selftest_value(p, sl, ph, SELFTEST_PANEL + i) == r0[2 * SELFTEST_PANEL + i])
									bad += " (owner rank " + std::to_string(p) + ": the value of phase " + (ph ? "B" : "A") + ", slot " + std::to_string(sl) + ")";
				}
			}
			for (long i = 0; i < SELFTEST_PANEL && bad.empty(); ++i)
				if (std::memcmp(&r0[i], &r0[SELFTEST_PANEL + i], 4) != 0)
					bad = "the W update's prologue (k_mu64_update32, PeerSlabs) read word " + std::to_string(i) + " of the ranks' panels differently from their contents (owner not determined: the r x r parts of the same buffers were read correctly)";
			if (!bad.empty()) bad = "round " + std::to_string(round) + " (pattern " + (phase ? "B" : "A") + ", slot " + std::to_string(slot) + "): " + bad;
		}
		for (int phase = 0; phase < 2 && bad.empty(); ++phase) {
			const float* cr = host.data() + 4 * res_per_round + (long)phase * (coll_count + (long)world * coll_count + coll_count);
			for (long i = 0; i < coll_count && bad.empty(); ++i) {
				float want = 0.f;
				for (int p = 0; p < world; ++p) want += selftest_value(p, 2, phase, i);
				if (cr[i] != want) bad = std::string("all-reduce (k_local_sum), pattern ") + (phase ? "B" : "A") + ", word " + std::to_string(i) + ": " + std::to_string(cr[i]) + " instead of " + std::to_string(want);
			}
			for (int p = 0; p < world && bad.empty(); ++p)
				for (long i = 0; i < coll_count && bad.empty(); ++i)
					if (cr[coll_count + (long)p * coll_count + i] != selftest_value(p, 3, phase, i))
						bad = std::string("all-gather (k_local_gather), pattern ") + (phase ? "B" : "A") + ": owner rank " + std::to_string(p) + ", word " + std::to_string(i) + ": " +
						      std::to_string(cr[coll_count + (long)p * coll_count + i]) + " instead of " + std::to_string(selftest_value(p, 3, phase, i));
			for (long i = 0; i < coll_count && bad.empty(); ++i) {
				float want = 0.f;
				for (int p = 0; p < world; ++p) want += selftest_value(p, 4, phase, (long)rank_ * coll_count + i);
				if (cr[coll_count + (long)world * coll_count + i] != want) bad = std::string("reduce-scatter (k_local_sum), pattern ") + (phase ? "B" : "A") + ", word " + std::to_string(i);
			}
		}
		if (!bad.empty()) {
			g.fail("peer-transport self-test: reader rank " + std::to_string(rank_) + " (device " + std::to_string(g.slots[rank_].device) + "): " + bad);
			g.aborted.store(true, std::memory_order_release);
			st = ST_HIP_ERROR;
		}
	} else if (!why.empty()) {
		g.fail("peer-transport self-test: rank " + std::to_string(rank_) + ": " + why);
		g.aborted.store(true, std::memory_order_release);
	}
	// nobody leaves (and frees what a peer may still read) before every rank's stream is through; then every rank learns the group's verdict
	if (s) (void)hipStreamSynchronize(s);
	(void)barrier_hard(g);
	const bool all_good = !g.aborted.load(std::memory_order_acquire);
	if (pool) (void)hipFree(pool);
	if (s) (void)hipStreamDestroy(s);
	(void)hipGetLastError();
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (report) {
		char txt[160];
		std::snprintf(txt, sizeof(txt), "peer-transport self-test: %d ranks, patterns A/B at the same addresses through the iteration's own kernels: %s (%.2f ms)", world, (all_good && st == ST_OK) ? "passed" : "FAILED", ms);
		*report = txt;
	}
	if (!all_good) return st != ST_OK ? st : ST_HIP_ERROR;
	return st;
}

Status local_comm_create(const std::shared_ptr<LocalGroup>& group, int rank, std::unique_ptr<Comm>* out) {
	if (!group || !out || rank < 0 || rank >= group->world) return ST_INVALID;
	LocalGroup::Slot& me = group->slots[rank];
	bool ok = hipGetDevice(&me.device) == hipSuccess &&
	          hipEventCreateWithFlags(&me.ready, hipEventDisableTiming) == hipSuccess &&
	          hipEventCreateWithFlags(&me.done, hipEventDisableTiming) == hipSuccess;
	if (!ok) { group->fail("rank " + std::to_string(rank) + ": no current device or no events"); group->aborted.store(true, std::memory_order_release); }
	if (!barrier(*group)) return ST_HIP_ERROR;
	// peers on other devices: map their memory into this device's address space (reads go over xGMI)
	for (int p = 0; p < group->world; ++p) {
		const int dev = group->slots[p].device;
		if (dev == me.device) continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, me.device, dev) != hipSuccess || !can) {
			group->fail("rank " + std::to_string(rank) + " (device " + std::to_string(me.device) + ") cannot map the memory of rank " + std::to_string(p) + " (device " + std::to_string(dev) +
			            "): hipDeviceCanAccessPeer says no");
			group->aborted.store(true, std::memory_order_release); break;
		}
		const hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
		if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
			group->fail("rank " + std::to_string(rank) + " (device " + std::to_string(me.device) + "): hipDeviceEnablePeerAccess(device " + std::to_string(dev) + ") failed: " + hipGetErrorString(e));
			group->aborted.store(true, std::memory_order_release); break;
		}
		(void)hipGetLastError();
	}
	if (!barrier(*group)) return ST_HIP_ERROR;
	std::unique_ptr<LocalComm> comm(new LocalComm(group, rank));
	// first contact of the ranks' devices, checked word by word before anything depends on it (~1 ms, once per communicator; also run when the ranks share a
	// device, where it can only fail for protocol reasons).  NMFAMD_SELFTEST=0 skips it.
	const char* force = std::getenv("NMFAMD_SELFTEST");
	if (group->world > 1 && !(force != nullptr && std::atoi(force) == 0)) {
		std::string report;
		const Status st = comm->self_test(&report);
		if (rank == 0) { std::lock_guard<std::mutex> l(group->failure_lock); group->selftest_report = report; }
		if (st != ST_OK) return st;
	}
	out->reset(comm.release());
	return ST_OK;
}

std::string local_group_selftest(LocalGroup& g) { std::lock_guard<std::mutex> l(g.failure_lock); return g.selftest_report; }

} // namespace nmfamd
