// runner.h -- what nmfgpu::compute's run / iteration loop (abi.cpp) drives: one engine on one GPU, or a team of
// rank threads with one column shard per GPU (internal header, included by abi.cpp only).
//
// The reference's dispatcher owns one IAlgorithm on one device (source/nmf/SingleGpuDispatcher.cpp:132-241).  Here the
// same loop talks to a Runner; with Parameter "numGpus" = N > 1 the runner is a TeamRunner: the calling thread is rank 0,
// N - 1 worker threads are the other ranks, every rank owns V(:, J_g), H(:, J_g), a replica of W and a communicator
// (sharded.h), and the whole factorisation stays inside the one blocking compute() call.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nmfgpu.h"
#include "comm.h"
#include "engine.h"
#include "host_init.h"
#include "sharded.h"

namespace nmfgpu {
namespace runner {

using nmfamd::Status;

template <typename T>
class Runner {
public:
	virtual ~Runner() {}
	// allocate device state and upload the input matrix
	virtual Status setup(const NmfDescription<T>& d) = 0;
	// start values of one run (d.seed holds the run's seed); want_h: the algorithm starts from H as well
	virtual Status init_run(NmfDescription<T>& d, bool want_h) = 0;
	virtual Status set_constant_w(NmfDescription<T>& d) = 0;
	virtual Status iterate(bool compute_error, bool constant_w) = 0;
	// The caller is about to wait for the error value of the iteration it has just enqueued and will, unless that value ends the run, enqueue another one:
	// whatever part of that next iteration changes nothing the caller could observe may be enqueued now (Engine::begin_next_iteration).  Optional.
	virtual Status begin_next_iteration() { return nmfamd::ST_OK; }
	virtual double frobenius() = 0;
	virtual double rmsd() = 0;
	virtual Status store(NmfDescription<T>& d) = 0;      // the factors into the caller's buffers
	virtual void synchronize() = 0;
	virtual const char* describe() const = 0;
	virtual const char* last_error() const { return ""; }
};

template <typename T>
Status upload_input(nmfamd::Engine<T>& engine, const MatrixDescription<T>& V) {
	switch (V.format) {
	case StorageFormat::Dense: return engine.upload_dense(V.dense.values, V.dense.leadingDimension);
	case StorageFormat::CSR: return engine.upload_sparse(1, V.csr.values, V.csr.rowPtr, V.csr.columnIndices, V.csr.nnz, V.csr.base == IndexBase::One ? 1 : 0);
	case StorageFormat::CSC: return engine.upload_sparse(2, V.csc.values, V.csc.columnPtr, V.csc.rowIndices, V.csc.nnz, V.csc.base == IndexBase::One ? 1 : 0);
	case StorageFormat::COO: return engine.upload_sparse(3, V.coo.values, V.coo.rowIndices, V.coo.columnIndices, V.coo.nnz, V.coo.base == IndexBase::One ? 1 : 0);
	}
	return nmfamd::ST_INVALID;
}

// Columns [col0, col0 + ncols) of the input matrix into a shard's engine.  Dense: a pointer offset.  CSR / CSC / COO ("sparse V shards by column
// blocks"): the rank filters the caller's arrays on the host into 0-based triplets of its block -- the index base applied to both index arrays exactly as
// the single-engine upload applies it (reference: cusparseSetMatIndexBase, Matrix.h:158-160,184-186,215-217), entries outside the matrix dropped,
// duplicates kept (they add) -- and uploads them as a COO block.
template <typename T>
Status upload_shard(nmfamd::Engine<T>& engine, const MatrixDescription<T>& V, long col0, long ncols) {
	if (V.format == StorageFormat::Dense) return engine.upload_dense(V.dense.values + (size_t)col0 * V.dense.leadingDimension, V.dense.leadingDimension);
	std::vector<int> ri, ci; std::vector<T> vals;
	const long m = V.rows, n = V.columns;
	auto push = [&](long i, long j, T v) { if (i >= 0 && i < m && j >= col0 && j < col0 + ncols) { ri.push_back((int)i); ci.push_back((int)(j - col0)); vals.push_back(v); } };
	switch (V.format) {
	case StorageFormat::CSR: {
		const long base = V.csr.base == IndexBase::One ? 1 : 0, nnz = V.csr.nnz;
		if (nnz > 0 && (!V.csr.values || !V.csr.rowPtr || !V.csr.columnIndices)) return nmfamd::ST_INVALID;
		for (long i = 0; i < m && nnz > 0; ++i)
			for (long p = (long)V.csr.rowPtr[i] - base; p < (long)V.csr.rowPtr[i + 1] - base && p < nnz; ++p) if (p >= 0) push(i, (long)V.csr.columnIndices[p] - base, V.csr.values[p]);
		break;
	}
	case StorageFormat::CSC: {
		const long base = V.csc.base == IndexBase::One ? 1 : 0, nnz = V.csc.nnz;
		if (nnz > 0 && (!V.csc.values || !V.csc.columnPtr || !V.csc.rowIndices)) return nmfamd::ST_INVALID;
		for (long j = col0; j < col0 + ncols && j < n && nnz > 0; ++j)
			for (long p = (long)V.csc.columnPtr[j] - base; p < (long)V.csc.columnPtr[j + 1] - base && p < nnz; ++p) if (p >= 0) push((long)V.csc.rowIndices[p] - base, j, V.csc.values[p]);
		break;
	}
	case StorageFormat::COO: {
		const long base = V.coo.base == IndexBase::One ? 1 : 0, nnz = V.coo.nnz;
		if (nnz > 0 && (!V.coo.values || !V.coo.rowIndices || !V.coo.columnIndices)) return nmfamd::ST_INVALID;
		for (long p = 0; p < nnz; ++p) push((long)V.coo.rowIndices[p] - base, (long)V.coo.columnIndices[p] - base, V.coo.values[p]);
		break;
	}
	default: return nmfamd::ST_INVALID;
	}
	return engine.upload_sparse(3, vals.data(), ri.data(), ci.data(), (long)vals.size(), 0);
}

// ---- one engine on the context's device ------------------------------------------------------------------------------
template <typename T>
class SingleRunner : public Runner<T> {
public:
	SingleRunner(const NmfDescription<T>& d, const nmfamd::AlgorithmParams& prm, hipStream_t stream) : prm_(prm), stream_(stream) { create(d); }
	Status setup(const NmfDescription<T>& d) override {
		if (Status st = engine_->allocate()) return st;
		Status st = upload_input(*engine_, d.inputMatrix);
		if (st == nmfamd::ST_VALUE_RANGE) {
			// infinities, NaN, |v| > 2^126 or 0 < |v| < 2^-100 in V: the split-operand product is not the fp32 product there;
			// take the native fp32 MFMA instructions (what Parameter "precision" = -1 selects)
			prm_.precision = -1;
			create(d);
			if (Status s2 = engine_->allocate()) return s2;
			st = upload_input(*engine_, d.inputMatrix);
		}
		return st;
	}
	// InitializationStrategy::create + initializeMatrixW/H (source/init/InitializationStrategy.cpp:36-47)
	Status init_run(NmfDescription<T>& d, bool want_h) override {
		const unsigned m = d.inputMatrix.rows, n = d.inputMatrix.columns, r = d.features;
		// (Parameter "nndsvd": the SVD-based start overrides initMethod -- host_init.cpp)
		switch (hostinit::nndsvd_variant(d.parameters, d.numParameters) >= 0 ? NmfInitializationMethod::MeanColumns : d.initMethod) {
		case NmfInitializationMethod::CopyExisting:
			return engine_->set_factors(d.outputMatrixW.dense.values, d.outputMatrixW.dense.leadingDimension,
			                           want_h ? d.outputMatrixH.dense.values : nullptr, d.outputMatrixH.dense.leadingDimension);
		case NmfInitializationMethod::AllRandomValues:
			return engine_->randomize_factors(d.seed, true, want_h);
		default: {
			// MeanColumns / k-means based strategies run on the host (north star: "init stays host-side C++")
			std::vector<T> W((size_t)m * r), H(want_h ? (size_t)r * n : 0);
			if (!hostinit::initialize<T>(d, W.data(), want_h ? H.data() : nullptr)) return nmfamd::ST_INVALID;
			return engine_->set_factors(W.data(), m, want_h ? H.data() : nullptr, r);
		}
		}
	}
	Status set_constant_w(NmfDescription<T>& d) override {
		return engine_->set_factors(d.outputMatrixW.dense.values, d.outputMatrixW.dense.leadingDimension, nullptr, 0);
	}
	Status iterate(bool compute_error, bool constant_w) override { return engine_->iterate(compute_error, constant_w); }
	Status begin_next_iteration() override { return engine_->begin_next_iteration(); }
	double frobenius() override { return engine_->frobenius(); }
	double rmsd() override { return engine_->rmsd(); }
	Status store(NmfDescription<T>& d) override {
		return engine_->get_factors(d.outputMatrixW.dense.values, d.outputMatrixW.dense.leadingDimension,
		                           d.outputMatrixH.dense.values, d.outputMatrixH.dense.leadingDimension);
	}
	void synchronize() override { (void)hipStreamSynchronize(engine_->stream()); }
	const char* describe() const override { return "one GPU"; }

private:
	void create(const NmfDescription<T>& d) {
		engine_.reset(new nmfamd::Engine<T>((int)d.inputMatrix.rows, (int)d.inputMatrix.columns, (int)d.features, static_cast<int>(d.algorithm), prm_));
		engine_->set_stream(stream_);
	}
	nmfamd::AlgorithmParams prm_;
	hipStream_t stream_;
	std::unique_ptr<nmfamd::Engine<T>> engine_;
};

// ---- N rank threads, one column shard each -----------------------------------------------------------------------------
// Transport: RCCL when every rank has a device of its own, the in-process peer-read transport when ranks share a device
// (more ranks than GPUs: one-GPU boxes, tests) or NMFAMD_COMM=p2p asks for it.
template <typename T>
class TeamRunner : public Runner<T> {
public:
	TeamRunner(const NmfDescription<T>& d, const nmfamd::AlgorithmParams& prm, int ranks, int first_device, int mode)
		: prm_(prm), world_(ranks), first_device_(first_device), mode_(mode), m_(d.inputMatrix.rows), n_(d.inputMatrix.columns), r_(d.features),
		  alg_(static_cast<int>(d.algorithm)) {}

	~TeamRunner() override {
		if (!workers_.empty()) {
			command(CMD_EXIT);
			for (std::thread& t : workers_) t.join();
		}
		// rank 0's objects are released here, on the thread (and device) that created them
		(void)hipSetDevice(first_device_);
		ranks_.clear();
	}

	Status setup(const NmfDescription<T>& d) override {
		int ndev = 0;
		if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return nmfamd::ST_NO_DEVICE; }
		if (world_ < 2 || world_ > 16 || (long)world_ > (long)n_) return nmfamd::ST_INVALID;
		// Transport: the in-process one (the ranks are threads of this process: every rank's kernels read the peers' buffers in place, over xGMI when the
		// devices differ) unless NMFAMD_COMM=rccl asks for RCCL; when two devices cannot map each other's memory the set-up falls back to RCCL (do_setup).
		// Round 4: RCCL was the default for ranks with a device each -- its mere presence costs ~11 us per iteration at config 2's size, and the
		// small-message form of the W step (sharded.cpp) needs peer reads.
		const char* force = std::getenv("NMFAMD_COMM");
		const bool shared = world_ > ndev;
		local_ = true;
		if (force != nullptr && std::strcmp(force, "rccl") == 0) { if (shared || !nmfamd::rccl_available()) return nmfamd::ST_INVALID; local_ = false; }
		rccl_fallback_ = local_ && !shared && nmfamd::rccl_available() && !(force != nullptr && (std::strcmp(force, "p2p") == 0 || std::strcmp(force, "local") == 0));
		for (int g = 0; g < world_; ++g) {
			ranks_.emplace_back(new Rank());
			ranks_[g]->device = (first_device_ + g) % ndev;
			nmfamd::shard_columns((long)n_, world_, g, &ranks_[g]->col0, &ranks_[g]->ncols);
		}
		rendezvous_ = nmfamd::local_group_create(world_);
		if (local_) transport_group_ = nmfamd::local_group_create(world_);
		else if (nmfamd::rccl_unique_id(unique_id_) != nmfamd::ST_OK) return nmfamd::ST_HIP_ERROR;
		input_ = &d;
		for (int g = 1; g < world_; ++g) workers_.emplace_back([this, g] { worker(g); });
		return command(CMD_SETUP);
	}

	Status init_run(NmfDescription<T>& d, bool want_h) override {
		init_method_ = d.initMethod; seed_ = d.seed; want_h_ = want_h;
		hostW_ = nullptr; hostH_ = nullptr; ldw_ = ldh_ = 0;
		std::vector<T> W, H;
		const bool svd_start = hostinit::nndsvd_variant(d.parameters, d.numParameters) >= 0;      // (Parameter "nndsvd" overrides initMethod)
		if (svd_start) init_method_ = NmfInitializationMethod::MeanColumns;                          // (any host-side method: the ranks take W and their columns of H)
		if (!svd_start && d.initMethod == NmfInitializationMethod::CopyExisting) {
			hostW_ = d.outputMatrixW.dense.values; ldw_ = d.outputMatrixW.dense.leadingDimension;
			hostH_ = d.outputMatrixH.dense.values; ldh_ = d.outputMatrixH.dense.leadingDimension;
		} else if (svd_start || d.initMethod != NmfInitializationMethod::AllRandomValues) {
			// the host-side initialisers see the whole matrix, once; every rank takes W and its columns of H
			W.resize((size_t)m_ * r_); H.resize(want_h ? (size_t)r_ * n_ : 0);
			if (!hostinit::initialize<T>(d, W.data(), want_h ? H.data() : nullptr)) return nmfamd::ST_INVALID;
			hostW_ = W.data(); ldw_ = m_; hostH_ = want_h ? H.data() : nullptr; ldh_ = r_;
		}
		return command(CMD_INIT);
	}
	// Constant basis vectors (every algorithm): W is given and stays, so the column shards are INDEPENDENT problems -- each rank fits its own columns of H
	// with its engine's single-GPU iteration (reference quirks of that mode included), no collective; ||V - W H||_F^2 is the sum of the shards' squares.
	Status set_constant_w(NmfDescription<T>& d) override {
		hostW_ = d.outputMatrixW.dense.values; ldw_ = d.outputMatrixW.dense.leadingDimension;
		// ALS / ACLS / AHCLS in this mode report the reference's literal term vector: tr(H^T W^T V) is replaced by sum_c ||W(:, c)||^2 (W traced against
		// itself).  That term is the same on every rank, so the shards' sums hold it N times where the one-GPU value holds it once.
		const_w_trace_ = 0.0;
		if (alg_ == nmfamd::ALG_ALS || alg_ == nmfamd::ALG_ACLS || alg_ == nmfamd::ALG_AHCLS) {
			for (long c = 0; c < r_; ++c)
				for (long i = 0; i < m_; ++i) { const double w = (double)hostW_[c * ldw_ + i]; const_w_trace_ += w * w; }
		}
		return command(CMD_CONSTW);
	}
	Status iterate(bool compute_error, bool constant_w) override {
		compute_error_ = compute_error;
		constant_w_ = constant_w;
		const Status st = command(CMD_ITERATE);
		if (st == nmfamd::ST_OK && constant_w && compute_error) return command(CMD_ERROR);
		return st;
	}
	double frobenius() override {
		if (constant_w_) { double s = 2.0 * (world_ - 1) * const_w_trace_; for (auto& rk : ranks_) s += rk->frob2; return std::sqrt(s); }
		return (!ranks_.empty() && ranks_[0]->sh) ? ranks_[0]->sh->frobenius() : 0.0;
	}
	double rmsd() override {
		if (constant_w_) return frobenius() / std::sqrt((double)(unsigned)((unsigned)m_ * (unsigned)n_));      // (unsigned product, like the reference)
		return (!ranks_.empty() && ranks_[0]->sh) ? ranks_[0]->sh->rmsd() : 0.0;
	}
	Status store(NmfDescription<T>& d) override {
		outW_ = d.outputMatrixW.dense.values; out_ldw_ = d.outputMatrixW.dense.leadingDimension;
		outH_ = d.outputMatrixH.dense.values; out_ldh_ = d.outputMatrixH.dense.leadingDimension;
		return command(CMD_STORE);
	}
	void synchronize() override { (void)command(CMD_SYNC); }
	const char* describe() const override {
		if (!local_) return "column shards, RCCL";
		describe_text_ = "column shards, in-process peer-read transport";
		if (transport_group_) { const std::string t = nmfamd::local_group_selftest(*transport_group_); if (!t.empty()) describe_text_ += "; " + t; }
		return describe_text_.c_str();
	}
	const char* last_error() const override { return error_.c_str(); }

private:
	enum Command { CMD_SETUP, CMD_INIT, CMD_ITERATE, CMD_STORE, CMD_SYNC, CMD_CONSTW, CMD_ERROR, CMD_EXIT };
	struct Rank {
		int device = 0;
		long col0 = 0, ncols = 0;
		hipStream_t stream = nullptr;
		std::unique_ptr<nmfamd::Engine<T>> eng;
		std::unique_ptr<nmfamd::Comm> comm;
		std::unique_ptr<nmfamd::ShardedRank<T>> sh;
		Status status = nmfamd::ST_OK;
		double frob2 = 0.0;                 // constant basis vectors: this shard's own squared Frobenius error
		~Rank() { sh.reset(); comm.reset(); eng.reset(); if (stream) (void)hipStreamDestroy(stream); }
	};

	// the calling thread is rank 0: publish the command, run rank 0's share, collect every rank's status
	Status command(Command c) {
		command_ = c;
		nmfamd::local_group_barrier(*rendezvous_);
		execute(0, c);
		nmfamd::local_group_barrier(*rendezvous_);
		for (const std::unique_ptr<Rank>& rk : ranks_)
			if (rk->status != nmfamd::ST_OK) {
				const char* a = rk->sh ? rk->sh->last_error() : "";
				const char* b = rk->eng ? rk->eng->last_error() : "";
				const char* c2 = rk->comm ? rk->comm->last_error() : "";
				error_ = std::string(a && *a ? a : (b && *b ? b : c2));
				if (error_.empty() && transport_group_) error_ = nmfamd::local_group_failure(*transport_group_);
				return rk->status;
			}
		return nmfamd::ST_OK;
	}
	void worker(int g) {
		for (;;) {
			nmfamd::local_group_barrier(*rendezvous_);
			const Command c = command_;
			if (c != CMD_EXIT) execute(g, c);
			else { Rank& rk = *ranks_[g]; (void)hipSetDevice(rk.device); rk.sh.reset(); rk.comm.reset(); rk.eng.reset(); if (rk.stream) { (void)hipStreamDestroy(rk.stream); rk.stream = nullptr; } }
			nmfamd::local_group_barrier(*rendezvous_);
			if (c == CMD_EXIT) return;
		}
	}
	void execute(int g, Command c) {
		Rank& rk = *ranks_[g];
		if (c == CMD_EXIT) return;
		if (rk.status != nmfamd::ST_OK && c != CMD_SETUP) {
			// a rank that failed keeps taking part in the rendezvous but does no more work; its peers' collectives
			// end through the transport's abort flag (in-process) -- with RCCL a failed rank is fatal for the call
			if (transport_group_) nmfamd::local_group_abort(*transport_group_);
			return;
		}
		switch (c) {
		case CMD_SETUP: rk.status = do_setup(g); break;
		case CMD_INIT: rk.status = rk.sh ? do_init(g) : nmfamd::ST_INVALID; break;      // (no sharded run: the set-up failed on some rank)
		case CMD_ITERATE: rk.status = !rk.sh ? nmfamd::ST_INVALID : constant_w_ ? rk.eng->iterate(compute_error_, true) : rk.sh->iterate(compute_error_); break;
		case CMD_CONSTW: rk.status = rk.sh ? rk.eng->set_factors(hostW_, ldw_, nullptr, 0) : nmfamd::ST_INVALID; break;
		case CMD_ERROR: rk.frob2 = rk.eng->frobenius_squared(); break;      // (the shard's own squared error: its columns of V against W H)
		case CMD_STORE: {
			T* hcols = outH_ + (size_t)rk.col0 * out_ldh_;
			rk.status = rk.sh ? rk.eng->get_factors(g == 0 ? outW_ : nullptr, out_ldw_, hcols, out_ldh_) : nmfamd::ST_INVALID;
			break;
		}
		case CMD_SYNC: if (rk.stream != nullptr && hipStreamSynchronize(rk.stream) != hipSuccess) rk.status = nmfamd::ST_HIP_ERROR; break;
		default: break;
		}
		// A rank that failed in the middle of a command leaves its peers inside a collective (spinning in the in-process transport's
		// publish / retire rendezvous): raise the abort flag NOW, before the closing rendezvous of the command, so that they return.
		// (RCCL has no such flag here: a rank failing between the collectives of an iteration is fatal for the process group;
		//  failures before the communicator exists are agreed upon in do_setup.)
		if (rk.status != nmfamd::ST_OK && transport_group_) nmfamd::local_group_abort(*transport_group_);
	}
	// Every rank runs every rendezvous of this function whatever happened to it (an early return would leave its peers waiting in a
	// barrier or in the communicator's own rendezvous), and the ranks AGREE before each step that is a collective: the communicator
	// is created only when every rank holds its engine and its shard, the sharded run is prepared (an all-gather) only when every
	// rank holds a communicator.
	Status do_setup(int g) {
		Rank& rk = *ranks_[g];
		Status st = nmfamd::ST_OK;
		if (hipSetDevice(rk.device) != hipSuccess) { (void)hipGetLastError(); st = nmfamd::ST_NO_DEVICE; }
		if (st == nmfamd::ST_OK && hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); rk.stream = nullptr; st = nmfamd::ST_HIP_ERROR; }
		const MatrixDescription<T>& V = input_->inputMatrix;
		auto make_engine = [&](const nmfamd::AlgorithmParams& prm) -> Status {
			rk.eng.reset(new nmfamd::Engine<T>((int)m_, (int)rk.ncols, (int)r_, alg_, prm));
			rk.eng->set_one_pass(false);       // rank threads may share a device: no persistent launch that claims every CU
			rk.eng->set_stream(rk.stream);
			if (mode_ == nmfamd::SHARD_ROW_BLOCKS) rk.eng->set_row_blocks(world_);
			Status s = rk.eng->allocate();
			if (s == nmfamd::ST_OK) s = upload_shard(*rk.eng, V, rk.col0, rk.ncols);
			return s;
		};
		if (st == nmfamd::ST_OK) st = make_engine(prm_);
		// values outside the exact range of the split-operand product in ANY shard: every rank switches to the native fp32
		// MFMA instructions (the ranks must run the same arithmetic: W is replicated)
		if (st == nmfamd::ST_VALUE_RANGE) odd_values_.store(true, std::memory_order_release);
		nmfamd::local_group_barrier(*rendezvous_);
		if (odd_values_.load(std::memory_order_acquire) && (st == nmfamd::ST_OK || st == nmfamd::ST_VALUE_RANGE)) {
			nmfamd::AlgorithmParams native = prm_;
			native.precision = -1;
			st = make_engine(native);
		}
		if (st != nmfamd::ST_OK) setup_failed_.store(true, std::memory_order_release);
		nmfamd::local_group_barrier(*rendezvous_);
		if (setup_failed_.load(std::memory_order_acquire)) return st;          // (a healthy rank reports ST_OK: the caller sees the failed one's status)
		Status ct = local_ ? nmfamd::local_comm_create(transport_group_, g, &rk.comm) : nmfamd::rccl_comm_create(unique_id_, world_, g, &rk.comm);
		if (ct != nmfamd::ST_OK) { comm_failed_.store(true, std::memory_order_release); if (transport_group_) nmfamd::local_group_abort(*transport_group_); }
		nmfamd::local_group_barrier(*rendezvous_);
		if (comm_failed_.load(std::memory_order_acquire) && local_ && rccl_fallback_) {
			// the in-process transport could not be set up (a pair of devices without peer access): every rank agrees (the failure aborted the group for
			// all), rank 0 fetches an RCCL id, and the ranks form an RCCL clique instead
			rk.comm.reset();
			if (g == 0) { fallback_id_ok_ = nmfamd::rccl_unique_id(unique_id_) == nmfamd::ST_OK; }
			nmfamd::local_group_barrier(*rendezvous_);
			ct = fallback_id_ok_ ? nmfamd::rccl_comm_create(unique_id_, world_, g, &rk.comm) : nmfamd::ST_HIP_ERROR;
			if (ct != nmfamd::ST_OK) setup_failed_.store(true, std::memory_order_release);
			nmfamd::local_group_barrier(*rendezvous_);
			if (g == 0 && !setup_failed_.load(std::memory_order_acquire)) { local_ = false; transport_group_.reset(); }
			nmfamd::local_group_barrier(*rendezvous_);
		} else if (comm_failed_.load(std::memory_order_acquire)) {
			setup_failed_.store(true, std::memory_order_release);
		}
		if (setup_failed_.load(std::memory_order_acquire)) return ct;
		rk.sh.reset(new nmfamd::ShardedRank<T>(rk.eng.get(), rk.comm.get(), mode_, (long)m_, (long)n_));
		return rk.sh->prepare();
	}
	Status do_init(int g) {
		Rank& rk = *ranks_[g];
		if (init_method_ == NmfInitializationMethod::AllRandomValues)
			return rk.eng->randomize_factors(seed_, true, want_h_, rk.col0);     // one stream for all shards
		const T* h = (want_h_ && hostH_ != nullptr) ? hostH_ + (size_t)rk.col0 * ldh_ : nullptr;
		return rk.eng->set_factors(hostW_, ldw_, h, ldh_);
	}

	nmfamd::AlgorithmParams prm_;
	int world_, first_device_, mode_;
	unsigned m_, n_, r_;
	int alg_;
	bool local_ = true;
	std::atomic<bool> odd_values_{false}, setup_failed_{false}, comm_failed_{false};
	bool rccl_fallback_ = false, fallback_id_ok_ = false;
	std::vector<std::unique_ptr<Rank>> ranks_;
	std::string error_;
	mutable std::string describe_text_;
	std::vector<std::thread> workers_;
	std::shared_ptr<nmfamd::LocalGroup> rendezvous_, transport_group_;
	unsigned char unique_id_[nmfamd::COMM_UNIQUE_ID_BYTES] = {0};
	// the command and its arguments (written by rank 0 before the rendezvous that releases the workers)
	Command command_ = CMD_SYNC;
	const NmfDescription<T>* input_ = nullptr;
	NmfInitializationMethod init_method_ = NmfInitializationMethod::CopyExisting;
	unsigned seed_ = 0;
	bool want_h_ = true, compute_error_ = false, constant_w_ = false;
	double const_w_trace_ = 0.0;
	const T* hostW_ = nullptr; const T* hostH_ = nullptr; long ldw_ = 0, ldh_ = 0;
	T* outW_ = nullptr; T* outH_ = nullptr; long out_ldw_ = 0, out_ldh_ = 0;
};

} // namespace runner
} // namespace nmfgpu
