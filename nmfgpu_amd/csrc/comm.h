// comm.h -- the collectives the column-sharded iteration needs (internal header).
//
// The reference is single-GPU (source/nmf/SingleGpuDispatcher.h:36); sharding is this engine's extension
// (SURVEY.md section 8e).  Two transports behind one interface:
//   * RcclComm  -- RCCL through its C API (librccl.so, loaded on first use; no torch anywhere): one rank per GPU,
//                  ranks may be threads of one process (nmfgpu::compute with Parameter "numGpus") or processes
//                  (bench.py under torch.distributed.run hands the unique id around);
//   * LocalComm -- ranks are threads of ONE process whose devices are the same GPU or peer-mapped GPUs: every rank's
//                  kernel reads the peers' buffers directly (over xGMI when the devices differ), sums in rank order.
//                  This is what runs when several ranks share a device (RCCL refuses that), e.g. on a one-GPU box.
// All operations are asynchronous on the caller's stream; counts are in elements of elem_bytes (4 or 8) bytes.
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace nmfamd {

class Comm {
public:
	virtual ~Comm() { for (void* p : xbuf_) if (p) (void)hipFree(p); }
	virtual int rank() const = 0;
	virtual int world() const = 0;
	virtual const char* transport() const = 0;
	// buf <- sum over ranks (in place), identical bits on every rank
	virtual Status all_reduce(void* buf, long count, int elem_bytes, hipStream_t s) = 0;
	// recv[0 .. count) <- sum over ranks p of send_p[rank * count .. (rank + 1) * count); send holds world * count elements
	virtual Status reduce_scatter(const void* send, void* recv, long count, int elem_bytes, hipStream_t s) = 0;
	// buf holds world * count elements; rank p's part [p * count, (p + 1) * count) is filled in from rank p (in place)
	virtual Status all_gather_inplace(void* buf, long count, int elem_bytes, hipStream_t s) = 0;
	// ---- exchange by direct reads (the small-message form of the W step, sharded.cpp) ------------------------------------------------------------
	// true: every rank can read every rank's exchange buffer where it lies (ranks are threads of one process on one device or on peer-mapped devices;
	// a team of one).  The consumer kernel then sums the ranks' buffers itself, in rank order -- no reduction kernel, no copy, ONE rendezvous per iteration.
	virtual bool direct_exchange() const { return world() == 1; }
	// `slots` buffers of `bytes` each (two: consecutive iterations alternate, which is what lets the readers of iteration k run while the writers of
	// iteration k + 1 already fill the other slot -- see exchange_publish).  The transport owns them.  mine[i] receives this rank's buffers.
	virtual Status exchange_alloc(size_t bytes, int slots, void** mine);
	// This rank has enqueued, on s, everything that writes its buffer `slot`.  On return s is ordered behind the writers of EVERY rank's buffer `slot`, and
	// peers[p] is rank p's buffer (p = 0 .. world - 1).  Slot reuse needs no second rendezvous: a rank overwrites slot i two publishes later, i.e. after it
	// waited (in the publish between) for every peer's NEXT writers, which those peers enqueued behind their readers of slot i.
	virtual Status exchange_publish(int slot, hipStream_t s, const void** peers);
	// several collectives issued between begin and end may be fused by the transport (ncclGroupStart / ncclGroupEnd)
	virtual void group_begin() {}
	virtual Status group_end() { return ST_OK; }
	virtual const char* last_error() const { return ""; }

protected:
	std::vector<void*> xbuf_;          // exchange buffers (this rank's)
	size_t xbuf_bytes_ = 0;
public:
	Comm() = default;
	Comm(const Comm&) = delete;
	Comm& operator=(const Comm&) = delete;
};

// ---- RCCL ---------------------------------------------------------------------------------------------------------
constexpr int COMM_UNIQUE_ID_BYTES = 128;       // sizeof(ncclUniqueId)
// false when librccl.so cannot be loaded (the text of the failure in *why)
bool rccl_available(const char** why = nullptr);
Status rccl_unique_id(void* out128);
// Collective over the ranks that share `id`: blocks until all `world` ranks have called it.  The calling thread's
// current HIP device is the rank's device.
Status rccl_comm_create(const void* id128, int world, int rank, std::unique_ptr<Comm>* out);

// ---- in-process -----------------------------------------------------------------------------------------------------
// Shared state of `world` rank threads.  Create one group, hand it to every rank thread, each calls local_comm_create
// (which blocks until all ranks have joined) with ITS device current.
struct LocalGroup;
std::shared_ptr<LocalGroup> local_group_create(int world);
Status local_comm_create(const std::shared_ptr<LocalGroup>& group, int rank, std::unique_ptr<Comm>* out);
// Host-side rendezvous of the group's rank threads (spins, then yields); and one word shared through it
void local_group_barrier(LocalGroup& g);
// every rank waiting in (or arriving at) a collective of this group returns an error instead of waiting on
void local_group_abort(LocalGroup& g);
// why a set-up on this group failed (which pair of devices could not map each other's memory); empty when nothing failed
std::string local_group_failure(LocalGroup& g);
// one line about the set-up self-test of the peer transport (empty: not run -- all ranks on one device and NMFAMD_SELFTEST unset)
std::string local_group_selftest(LocalGroup& g);

} // namespace nmfamd
