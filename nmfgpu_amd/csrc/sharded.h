// sharded.h -- one rank of the column-sharded multiplicative update (internal header).
//
// Rank g holds V(:, J_g), H(:, J_g) and a replica of W.  Per iteration:
//     H step                      local (W^T W, W^T V_g, update of H(:, J_g)): no communication
//     exchange <- [ (V_g H_g^T)^T | H_g H_g^T ]     (nsNMF: of the smoothed S H_g)
//   mode ROW_BLOCKS (default):
//     reduce-scatter of the panel by row blocks of W + all-reduce of H H^T (r x r)
//     every rank updates ITS m / N rows of W;  all-reduce of the r column sums of squares;  normalise;
//     all-gather of the row blocks  ->  every rank holds the same bits of the new W
//   mode REPLICATED:
//     all-reduce of the whole exchange buffer, every rank applies the identical W update (the round-1 scheme)
// Error iterations: the per-column terms of tr(H^T W^T V) are gathered so that the sorted host summation
// (source/nmf/FrobeniusResolver.cpp:29-51) sees the single-GPU vectors; nothing waits for the GPU until somebody
// reads the error.
#pragma once

#include <algorithm>

#include <vector>

#include "comm.h"
#include "engine.h"

namespace nmfamd {

enum ShardMode { SHARD_ROW_BLOCKS = 0, SHARD_REPLICATED = 1 };

// columns [first, first + count) of rank `rank` when `total` columns are dealt to `world` ranks
inline void shard_columns(long total, int world, int rank, long* first, long* count) {
	const long a = (total * rank) / world, b = (total * (rank + 1)) / world;
	*first = a; *count = b - a;
}

template <typename T>
class ShardedRank {
public:
	// engine: this rank's shard (created with set_row_blocks(comm->world()) when mode is SHARD_ROW_BLOCKS); not owned
	ShardedRank(Engine<T>* engine, Comm* comm, int mode, long rows, long total_columns);
	~ShardedRank();
	Status prepare();                       // buffers + the gathered, sorted tr(V^T V) terms (after the upload)
	Status iterate(bool compute_error);
	Status run(int count, int first_iteration, int error_every, int last_iteration);
	// row-block mode with the fragment exchange: the fp32 rows of every rank's block into the engine's W panel -- a COLLECTIVE (nmfamd_sharded_gather_w)
	Status gather_w_rows();
	// (KL update: the engine resolves its own, reduced, error terms -- per row of W, not per column)
	double frobenius() { if (eng_->is_kl()) return eng_->frobenius(); finalize(); return frob_; }
	double rmsd() { if (eng_->is_kl()) return eng_->rmsd(); finalize(); return rmsd_; }
	double kl_divergence() { return eng_->is_kl() ? eng_->kl_divergence() : 0.0; }
	int mode() const { return mode_; }
	const char* last_error() const { return last_error_; }

private:
	Status fail(const char* what) { last_error_ = what; (void)hipGetLastError(); return ST_HIP_ERROR; }
	Status launch_error_gather();
	// elements per rank in the gathered error-term buffer: [n_local terms | r terms], padded to whole 16-byte units
	long slot_len() const { return ((std::max<long>(nloc_max_, eng_->r()) + eng_->r() + 3) / 4) * 4; }
	void finalize();

	Engine<T>* eng_;
	Comm* comm_;
	int mode_;
	long rows_, total_columns_;
	long blk_rows_ = 0;                     // rows of W per rank (ROW_BLOCKS)
	long nloc_max_ = 0;
	T* exchange_ = nullptr;                 // [panel RP x mpad | H H^T RP x RP]
	bool direct_ = false, rehearse_ = false; // the W update reads the ranks' exchange buffers itself (sharded.cpp, prepare)
	bool rehearse_rows_ = false;             // measurement build: a team of one updates 1 / N of the rows (timing only)
	T* xslot_[2] = {nullptr, nullptr};      // ... this rank's two exchange buffers (the transport's), alternating by iteration
	unsigned long iterations_ = 0;
	T* blk_ = nullptr;                      // reduced (V H^T)^T rows of this rank
	T* colsq_ = nullptr;                    // RP sums of squares
	T* err_dev_ = nullptr;                  // world * (nloc_max + r) gathered error terms
	T* err_pin_ = nullptr;
	T* err_pin_dev_ = nullptr;      // the device's address of err_pin_
	hipEvent_t err_event_ = nullptr;
	bool err_pending_ = false;
	std::vector<T> vtv_all_;
	double frob_ = 0, rmsd_ = 0;
	const char* last_error_ = "";
};

} // namespace nmfamd
