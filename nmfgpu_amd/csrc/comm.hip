// comm.hip -- RCCL (C API, loaded at run time) and in-process transports of comm.h.
#include "comm.h"

#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
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
	void fail(const std::string& what) { std::lock_guard<std::mutex> g(failure_lock); if (failure.empty()) failure = what; }
	struct Slot {
		const void* buf = nullptr;
		hipEvent_t ready = nullptr, done = nullptr;
		int device = -1;
		// exchange by direct reads (Comm::exchange_alloc / exchange_publish): this rank's two buffers and the events behind their writers
		const void* xbuf[2] = {nullptr, nullptr};
		hipEvent_t xready[2] = {nullptr, nullptr};
	};
	std::vector<Slot> slots;
	~LocalGroup() {
		for (Slot& s : slots) {
			if (s.ready) (void)hipEventDestroy(s.ready);
			if (s.done) (void)hipEventDestroy(s.done);
			for (hipEvent_t e : s.xready) if (e) (void)hipEventDestroy(e);
		}
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

class LocalComm : public Comm {
public:
	LocalComm(std::shared_ptr<LocalGroup> g, int rank) : g_(std::move(g)), rank_(rank) {}
	~LocalComm() override { if (scratch_) (void)hipFree(scratch_); }
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
		Status st = Comm::exchange_alloc(bytes, slots, mine);
		LocalGroup::Slot& me = g_->slots[rank_];
		for (int i = 0; i < slots && st == ST_OK; ++i) {
			me.xbuf[i] = mine[i];
			if (me.xready[i] == nullptr && hipEventCreateWithFlags(&me.xready[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); st = ST_HIP_ERROR; }
		}
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
	out->reset(new LocalComm(group, rank));
	return ST_OK;
}

} // namespace nmfamd
