// sharded.cpp -- one rank of the column-sharded multiplicative update (see sharded.h).
#include "sharded.h"
#include "tuning.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace nmfamd {

template <typename T>
ShardedRank<T>::ShardedRank(Engine<T>* engine, Comm* comm, int mode, long rows, long total_columns)
	: eng_(engine), comm_(comm), mode_(mode), rows_(rows), total_columns_(total_columns) {}

template <typename T>
ShardedRank<T>::~ShardedRank() {
	if (eng_) eng_->set_w_gather_hook(nullptr);       // (an engine that outlives its run and still lacks the other ranks' rows says so instead of handing out stale ones)
	if (exchange_) (void)hipFree(exchange_);
	if (blk_) (void)hipFree(blk_);
	if (colsq_) (void)hipFree(colsq_);
	if (err_dev_) (void)hipFree(err_dev_);
	if (err_pin_) (void)hipHostFree(err_pin_);
	if (err_event_) (void)hipEventDestroy(err_event_);
}

template <typename T>
Status ShardedRank<T>::prepare() {
	if (!eng_ || !comm_ || (mode_ != SHARD_ROW_BLOCKS && mode_ != SHARD_REPLICATED)) return ST_INVALID;
	if (eng_->is_kl() && mode_ != SHARD_REPLICATED) { last_error_ = "KL update: the sharded W step is the replicated form (shard mode 1)"; return ST_INVALID; }
	if (eng_->error_terms_per_factor_row() && mode_ != SHARD_REPLICATED) { last_error_ = "GDCLS / ALS family: the sharded W step is the replicated form (shard mode 1)"; return ST_INVALID; }
	const int world = comm_->world(), rank = comm_->rank();
	// The small-message form of the W step (replicated mode, rank-64 multiplicative update on the split-operand path, a transport whose ranks can read each
	// other's memory): no reduction kernel and no copy -- the W update reads every rank's exchange panel where it lies (Engine::w_finish_peers), two exchange
	// buffers alternate, ONE rendezvous per iteration (Comm::exchange_publish).  NMFAMD_SHARD_REHEARSE=1 makes a team of ONE go through exactly that path
	// (panel written, published, read back by pointer, r x r part summed by the small launch) instead of the fused single-GPU iteration it would otherwise
	// equal: what a rank of an N-GPU team runs, minus link time and the cross-device event wait -- the basis of DESIGN section 6's projection.
	const char* rehearse = tuning_env("NMFAMD_SHARD_REHEARSE");
	rehearse_ = world == 1 && rehearse != nullptr && std::atoi(rehearse) != 0;
	const char* no_direct = tuning_env("NMFAMD_SHARD_NO_DIRECT");
	direct_ = mode_ == SHARD_REPLICATED && eng_->direct_w_finish() && comm_->direct_exchange() && !(no_direct != nullptr && std::atoi(no_direct) != 0);
	eng_->set_sole_rank(world == 1 && mode_ == SHARD_REPLICATED && !rehearse_);
	long first = 0, count = 0;
	shard_columns(total_columns_, world, rank, &first, &count);
	if (count != eng_->n() || rows_ != eng_->m()) { last_error_ = "shard shape does not match shard_columns()"; return ST_INVALID; }
	const long RP = eng_->rp(), mpad = eng_->mpad();
	hipStream_t s = eng_->stream();
	auto dalloc = [&](T** p, long elems) -> bool {
		if (hipMalloc((void**)p, sizeof(T) * (size_t)elems) != hipSuccess) return false;
		return hipMemsetAsync(*p, 0, sizeof(T) * (size_t)elems, s) == hipSuccess;
	};
	if (direct_) {
		void* mine[2] = {nullptr, nullptr};
		if (Status st = comm_->exchange_alloc(sizeof(T) * (size_t)eng_->exchange_count(), 2, mine)) { last_error_ = "exchange_alloc"; return st; }
		xslot_[0] = static_cast<T*>(mine[0]); xslot_[1] = static_cast<T*>(mine[1]);
	} else if (!dalloc(&exchange_, eng_->exchange_count())) return fail("hipMalloc(exchange)");
	if (mode_ == SHARD_ROW_BLOCKS) {
		// (measurements: NMFAMD_SHARD_REHEARSE = N > 1 on a team of one in this mode makes the rank run the row-block W step on 1 / N of the rows, as a rank of N would:
		//  timing only -- the other row blocks are never updated, the factors mean nothing.  tools/c4_shard_modes.py)
		const int pretend = (world == 1 && rehearse != nullptr && std::atoi(rehearse) > 1) ? std::atoi(rehearse) : world;
		if (mpad % (128l * pretend) != 0) { last_error_ = "engine was not created with set_row_blocks(world)"; return ST_INVALID; }
		blk_rows_ = mpad / pretend;
		rehearse_rows_ = pretend != world;
		eng_->set_w_gather_hook([this]() { return gather_w_rows(); });
		if (!dalloc(&blk_, RP * blk_rows_) || !dalloc(&colsq_, RP)) return fail("hipMalloc(row block)");
	}
	for (int p = 0; p < world; ++p) { long f, c; shard_columns(total_columns_, world, p, &f, &c); nloc_max_ = std::max(nloc_max_, c); }
	const long L = slot_len();
	if (!dalloc(&err_dev_, L * world)) return fail("hipMalloc(error terms)");
	if (hipHostMalloc((void**)&err_pin_, sizeof(T) * (size_t)(L * world)) != hipSuccess) return fail("hipHostMalloc(error terms)");
	{
		void* dp = nullptr;
		if (hipHostGetDevicePointer(&dp, err_pin_, 0) == hipSuccess) err_pin_dev_ = static_cast<T*>(dp); else (void)hipGetLastError();
	}
	if (hipEventCreateWithFlags(&err_event_, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");

	// the sorted tr(V^T V) terms of ALL columns, once (V does not change): gather the local vectors through the
	// error-term buffer, sort the concatenation on the host
	const std::vector<T>& local = eng_->terms_vtv_sorted();
	if ((long)local.size() != count) { last_error_ = "upload V before prepare()"; return ST_INVALID; }
	if (count > 0 && hipMemcpyAsync(err_dev_ + (long)rank * L, local.data(), sizeof(T) * (size_t)count, hipMemcpyHostToDevice, s) != hipSuccess) return fail("hipMemcpyAsync(vtv)");
	if (Status st = comm_->all_gather_inplace(err_dev_, L, (int)sizeof(T), s)) { last_error_ = comm_->last_error(); return st; }
	if (hipMemcpyAsync(err_pin_, err_dev_, sizeof(T) * (size_t)(L * world), hipMemcpyDeviceToHost, s) != hipSuccess) return fail("hipMemcpyAsync(vtv back)");
	if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize");
	vtv_all_.clear();
	for (int p = 0; p < world; ++p) { long f, c; shard_columns(total_columns_, world, p, &f, &c); vtv_all_.insert(vtv_all_.end(), err_pin_ + (long)p * L, err_pin_ + (long)p * L + c); }
	std::sort(vtv_all_.begin(), vtv_all_.end());
	if (eng_->is_kl()) {
		// the KL divergence needs sum(V) over ALL columns: every rank's sum as a (high, low) pair of T, summed over the ranks (a plain fp32 sum of
		// 10^7-sized values would lose the units digit)
		const double mine = eng_->sum_v();
		T pair[4] = {(T)mine, (T)(mine - (double)(T)mine), T(0), T(0)};
		if (hipMemcpyAsync(err_dev_, pair, sizeof(T) * 4, hipMemcpyHostToDevice, s) != hipSuccess) return fail("hipMemcpyAsync(sum V)");
		if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize");
		if (world > 1) { if (Status st = comm_->all_reduce(err_dev_, 4, (int)sizeof(T), s)) { last_error_ = comm_->last_error(); return st; } }
		if (hipMemcpyAsync(pair, err_dev_, sizeof(T) * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("hipMemcpyAsync(sum V back)");
		if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize");
		eng_->set_error_globals(vtv_all_, (double)pair[0] + (double)pair[1], total_columns_);
	}
	if (hipMemsetAsync(err_dev_, 0, sizeof(T) * (size_t)(L * world), s) != hipSuccess) return fail("hipMemsetAsync");
	return ST_OK;
}

template <typename T>
Status ShardedRank<T>::iterate(bool compute_error) {
	const int world = comm_->world(), rank = comm_->rank();
	const long RP = eng_->rp(), mpad = eng_->mpad();
	const int eb = (int)sizeof(T);
	hipStream_t s = eng_->stream();
	auto comm_fail = [&](Status st) { last_error_ = comm_->last_error(); return st; };
	// this run gathers the error terms of all ranks on the device (launch_error_gather) and resolves them itself: the engine's own copy to the host and its
	// event are skipped while an iteration of the run is being enqueued (KL: the terms travel in the exchange buffer and the engine keeps them)
	struct StayGuard { Engine<T>* e; bool on; ~StayGuard() { if (on) e->set_error_terms_stay_on_device(false); } } stay{eng_, !eng_->is_kl()};
	if (stay.on) eng_->set_error_terms_stay_on_device(true);
	if (Status st = eng_->h_step(compute_error)) { last_error_ = eng_->last_error(); return st; }
	if (direct_) {
		T* mine = xslot_[iterations_++ & 1];
		if (Status st = eng_->w_products(mine)) { last_error_ = eng_->last_error(); return st; }
		if (world == 1 && !rehearse_) {
			if (Status st = eng_->w_finish(mine, compute_error)) { last_error_ = eng_->last_error(); return st; }
		} else {
			const void* peers[PEER_SLABS_MAX];
			if (world > PEER_SLABS_MAX) return ST_INVALID;
			if (Status st = comm_->exchange_publish((int)((iterations_ - 1) & 1), s, peers)) return comm_fail(st);
			if (Status st = eng_->w_finish_peers(reinterpret_cast<const T* const*>(peers), world, compute_error)) { last_error_ = eng_->last_error(); return st; }
		}
		if (compute_error) return launch_error_gather();
		return ST_OK;
	}
	if (Status st = eng_->w_products(exchange_)) { last_error_ = eng_->last_error(); return st; }
	if (mode_ == SHARD_REPLICATED) {
		if (world > 1) { if (Status st = comm_->all_reduce(exchange_, eng_->exchange_count(), eb, s)) return comm_fail(st); }
		if (Status st = eng_->w_finish(exchange_, compute_error)) { last_error_ = eng_->last_error(); return st; }
	} else {
		T* hht = exchange_ + RP * mpad;
		const long row0 = (long)rank * blk_rows_;
		// m x r sums by row blocks (every link carries 1/N of the panel), the r x r sums to everybody
		comm_->group_begin();
		Status a = comm_->reduce_scatter(exchange_, blk_, RP * blk_rows_, eb, s);
		Status b = a == ST_OK ? comm_->all_reduce(hht, RP * RP, eb, s) : a;
		Status c = comm_->group_end();
		if (a != ST_OK || b != ST_OK || c != ST_OK) return comm_fail(a != ST_OK ? a : (b != ST_OK ? b : c));
		if (Status st = eng_->w_update_rows(blk_, hht, row0, blk_rows_, compute_error, colsq_)) { last_error_ = eng_->last_error(); return st; }
		if (Status st = comm_->all_reduce(colsq_, RP, eb, s)) return comm_fail(st);
		if (Status st = eng_->w_normalize_rows(row0, blk_rows_, colsq_)) { last_error_ = eng_->last_error(); return st; }
		if (eng_->w_fragment_exchange() && blk_rows_ % 16 == 0) {
			// exchange what the consumers read (round 5): until the next W update the other ranks need this rank's rows only as the bf16 fragments that
			// w_normalize_rows() just wrote -- half the bytes of the fp32 rows (config 4: 25.6 MB over the links instead of 51.2), and no rank re-rounds all rows
			// (25 us); the fp32 rows are gathered when somebody asks for the factors (gather_w_rows, through the engine's hook)
			if (Status st = comm_->all_gather_inplace(eng_->w_fragments(), eng_->w_fragment_words_per_row() * blk_rows_, 4, s)) return comm_fail(st);
			eng_->w_fragments_gathered(world > 1 || rehearse_rows_);
		} else {
			if (Status st = comm_->all_gather_inplace(eng_->w_panel(), RP * blk_rows_, eb, s)) return comm_fail(st);
			eng_->w_rows_replaced();
		}
	}
	if (compute_error && !eng_->is_kl()) return launch_error_gather();      // (KL: the terms travelled in the exchange buffer, the engine keeps them)
	return ST_OK;
}

// The fp32 rows of every rank's block into w_panel() (row-block mode with the fragment exchange: Engine::get_factors and friends call this through the hook).
// A COLLECTIVE: every rank's engine must ask for its factors.
template <typename T>
Status ShardedRank<T>::gather_w_rows() {
	if (mode_ != SHARD_ROW_BLOCKS || blk_rows_ <= 0) return ST_OK;
	if (comm_->world() > 1) {
		if (Status st = comm_->all_gather_inplace(eng_->w_panel(), eng_->rp() * blk_rows_, (int)sizeof(T), eng_->stream())) { last_error_ = comm_->last_error(); return st; }
	}
	eng_->w_rows_gathered();
	return ST_OK;
}

template <typename T>
Status ShardedRank<T>::launch_error_gather() {
	finalize();        // the previous error iteration's terms arrived long ago; the landing buffer is reused
	const int world = comm_->world(), rank = comm_->rank();
	const long L = slot_len();
	hipStream_t s = eng_->stream();
	if (world == 1 && !rehearse_ && err_pin_dev_ != nullptr) {
		// a team of one: nothing to gather -- the terms go straight to the pinned buffer
		if (eng_->error_terms_to_device(err_pin_dev_, L) < 0) return fail("error_terms_to_device");
		if (hipEventRecord(err_event_, s) != hipSuccess) return fail("hipEventRecord");
		err_pending_ = true;
		return ST_OK;
	}
	if (eng_->error_terms_to_device(err_dev_ + (long)rank * L, L) < 0) return fail("error_terms_to_device");
	if (Status st = comm_->all_gather_inplace(err_dev_, L, (int)sizeof(T), s)) { last_error_ = comm_->last_error(); return st; }
	// (a kernel that writes the pinned buffer: the runtime's device-to-host copy idles the stream ~18 us around its blit -- Engine::fetch_error_terms)
	if (err_pin_dev_ != nullptr) { if (launch_copy_small<T>(err_pin_dev_, err_dev_, L * world, s) != hipSuccess) return fail("copy of the error terms"); }
	else if (hipMemcpyAsync(err_pin_, err_dev_, sizeof(T) * (size_t)(L * world), hipMemcpyDeviceToHost, s) != hipSuccess) return fail("hipMemcpyAsync(error terms)");
	if (hipEventRecord(err_event_, s) != hipSuccess) return fail("hipEventRecord");
	err_pending_ = true;
	return ST_OK;
}

template <typename T>
void ShardedRank<T>::finalize() {
	if (!err_pending_) return;
	err_pending_ = false;
	(void)hipEventSynchronize(err_event_);
	const int world = comm_->world(), rank = comm_->rank();
	const long L = slot_len();
	std::vector<T> htwtv, hhtwtw;
	if (eng_->error_terms_per_factor_row()) {
		// GDCLS / ALS family: r terms of tr(H^T W^T V) and r of tr(H H^T W^T W), both from the reduced sums: this rank's own slot
		const T* mine = err_pin_ + (long)rank * L;
		htwtv.assign(mine, mine + eng_->r());
		hhtwtw.assign(mine + eng_->r(), mine + 2 * eng_->r());
	} else {
		for (int p = 0; p < world; ++p) { long f, c; shard_columns(total_columns_, world, p, &f, &c); htwtv.insert(htwtv.end(), err_pin_ + (long)p * L, err_pin_ + (long)p * L + c); }
		// the r terms of tr(H H^T W^T W) come from the reduced H H^T and the replicated W^T W: identical on every rank
		const T* mine = err_pin_ + (long)rank * L + eng_->n();
		hhtwtw.assign(mine, mine + eng_->r());
	}
	frob_ = resolve_frobenius<T>(vtv_all_, htwtv, hhtwtw);
	rmsd_ = frob_ / std::sqrt((double)(unsigned)((unsigned)rows_ * (unsigned)total_columns_));   // (unsigned product, like the reference)
}

template <typename T>
Status ShardedRank<T>::run(int count, int first_iteration, int error_every, int last_iteration) {
	for (int k = 0; k < count; ++k) {
		const int it = first_iteration + k;
		const bool err = (error_every > 0 && it % error_every == 0) || (last_iteration > 0 && it == last_iteration);
		if (Status st = iterate(err)) return st;
	}
	// the batch that ends the run (every rank is told the same last_iteration): the fp32 rows of the other ranks' blocks are gathered HERE, inside the collective
	// call, so that a single rank asking for its factors afterwards (nmfamd_engine_get_factors) neither starts a collective alone nor finds the hook gone
	// after nmfamd_sharded_destroy
	if (last_iteration > 0 && count > 0 && first_iteration + count - 1 == last_iteration && eng_->w_rows_stale()) return gather_w_rows();
	return ST_OK;
}

template class ShardedRank<float>;
template class ShardedRank<double>;

} // namespace nmfamd
