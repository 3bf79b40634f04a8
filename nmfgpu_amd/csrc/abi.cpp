// abi.cpp -- the entry points of include/nmfgpu.h on top of the MI355X engine.
//
// Replaces source/common/Interface.cpp (API bodies), source/nmf/SingleGpuDispatcher.cpp (run and
// iteration loop), source/nmf/Summary.cpp and source/common/Logging.cpp of the reference.  The
// observable behaviour kept from there, with the reference line it comes from:
//   * initialize() is per thread; a second call returns ErrorAlreadyInitialized   Interface.cpp:53-66
//   * compute() before initialize() -> ErrorNotInitialized                        :216-218
//   * CopyExisting with numRuns > 1 warns and clamps numRuns (caller's struct)     :221-225
//   * features > columns without constant basis vectors -> ErrorInvalidArgument    :228-232
//   * required Parameter names per algorithm, looked up by strcmp                  :41-49, :249-326
//   * run loop, error every 10th / last iteration, |delta| < threshold stop,
//     best-run store, interrupt poll once per iteration         SingleGpuDispatcher.cpp:155-235
//   * description.seed <- next draw of mt19937(seed) before every run              Algorithm.cpp:26-31
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <vector>

#include "../../include/nmfgpu.h"
#include "engine.h"
#include "host_init.h"
#include "runner.h"

namespace nmfgpu {
namespace {

// ---- logging (source/common/Logging.h:89-109: one process-wide verbosity) ------------------
std::atomic<Verbosity> g_verbosity{Verbosity::Summary};

bool allowed(Verbosity level) { return static_cast<int>(g_verbosity.load(std::memory_order_relaxed)) >= static_cast<int>(level); }

void log_error(const char* text) { std::cerr << text << std::endl; }
void log_summary(const char* text) { if (allowed(Verbosity::Summary)) { std::cout << text; std::cout.flush(); } }

// ---- per-thread context (Interface.cpp:51; the vendor handles become one HIP stream) --------
struct DeviceContext {
	int deviceID = 0;
	hipStream_t stream = nullptr;
	bool stream_tried = false;
};
thread_local DeviceContext* g_context = nullptr;

bool ensure_stream(DeviceContext& ctx) {
	if (ctx.stream) return true;
	if (ctx.stream_tried) return false;
	ctx.stream_tried = true;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return false; }
	if (hipSetDevice(ctx.deviceID) != hipSuccess) { (void)hipGetLastError(); return false; }
	if (hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx.stream = nullptr; return false; }
	return true;
}

void drop_stream(DeviceContext& ctx) {
	if (ctx.stream) (void)hipStreamDestroy(ctx.stream);
	ctx.stream = nullptr;
	ctx.stream_tried = false;
}

// ---- Summary (source/nmf/Summary.cpp:27-60) -------------------------------------------------
class SummaryImpl : public ISummary {
	std::vector<ExecutionRecord> records_;
	unsigned best_ = 0;
public:
	void destroy() override { delete this; }
	unsigned bestRun() const override { return best_; }
	void record(unsigned index, ExecutionRecord& out) const override { if (index < records_.size()) out = records_[index]; }
	unsigned recordCount() const override { return unsigned(records_.size()); }
	// a record is the new best only if it is STRICTLY below every earlier one (:47-55)
	void insert(const ExecutionRecord& rec) {
		records_.push_back(rec);
		for (size_t i = 0; i + 1 < records_.size(); ++i)
			if (records_[i].frobenius <= rec.frobenius) return;
		best_ = unsigned(records_.size() - 1);
	}
	void reset() { best_ = 0; records_.clear(); }
};

int parameter_index(const Parameter* parameters, unsigned count, const char* name) {
	for (unsigned i = 0; i < count; ++i)
		if (parameters[i].name && std::strcmp(parameters[i].name, name) == 0) return int(i);
	return -1;
}

// ---- progress table (SingleGpuDispatcher.cpp:44-130) ----------------------------------------
void duration_string(char (&buffer)[32], long long ms) {
	long long h = ms / 3600000; ms %= 3600000;
	long long m = ms / 60000; ms %= 60000;
	std::snprintf(buffer, sizeof(buffer), "%02d:%02d:%02d.%03d", int(h), int(m), int(ms / 1000), int(ms % 1000));
}

void print_header(bool multi) {
	if (!allowed(Verbosity::Summary)) return;
	const char* rule = multi ? " -------------------------------------------------------------------------------------------------------------\n"
	                         : " ---------------------------------------------------------------------------------------------------\n";
	log_summary(rule);
	log_summary(multi ? " |   Run   | Iteration |     Frobenius     |       RMSD       |       Delta      | Elapsed Time |   Status   |\n"
	                  : " | Iteration |     Frobenius     |       RMSD       |       Delta      | Elapsed Time |   Status   |\n");
	log_summary(rule);
}

void print_row(bool multi, bool final_row, unsigned run, unsigned runs, unsigned iteration, double frob, double rmsd, double delta, long long ms, const char* status) {
	if (!allowed(Verbosity::Summary)) return;
	char tbuf[32]; duration_string(tbuf, ms);
	char marquee[16] = "          ";
	if (!final_row) { unsigned i = iteration / 10; marquee[i % 10] = '<'; marquee[(i + 1) % 10] = '='; marquee[(i + 2) % 10] = '>'; status = marquee; }
	char line[512];
	const char* lead = (final_row || iteration != 10) ? "\r" : "";
	if (multi) std::snprintf(line, sizeof(line), "%s | %7u | %9u | %17.4f | %16.4f | %16.4f | %s | %10s |%s", lead, run, iteration, frob, rmsd, delta, tbuf, status, final_row ? "\n" : "");
	else std::snprintf(line, sizeof(line), "%s | %9u | %17.4f | %16.4f | %16.4f | %s | %10s |%s", lead, iteration, frob, rmsd, delta, tbuf, status, final_row ? "\n" : "");
	log_summary(line);
	if (final_row && (!multi || run == runs))
		log_summary(multi ? " -------------------------------------------------------------------------------------------------------------\n"
		                  : " ---------------------------------------------------------------------------------------------------\n");
}

const char* algorithm_name(NmfAlgorithm a) {
	switch (a) {
	case NmfAlgorithm::Multiplicative: return "Multiplicative Frobenius";
	case NmfAlgorithm::GDCLS: return "Gradient Descent Constrained Least Squares";
	case NmfAlgorithm::ALS: return "Alternating Least Squares";
	case NmfAlgorithm::ACLS: return "Alternating Constrained Least Squares";
	case NmfAlgorithm::AHCLS: return "Alternating Hoyer Constrained Least Squares";
	case NmfAlgorithm::nsNMF: return "non-smooth NMF";
	}
	return "?";
}

ResultType from_status(nmfamd::Status s) {
	switch (s) {
	case nmfamd::ST_OK: return ResultType::Success;
	case nmfamd::ST_INVALID: return ResultType::ErrorInvalidArgument;
	case nmfamd::ST_NO_DEVICE_MEMORY: return ResultType::ErrorNotEnoughDeviceMemory;
	case nmfamd::ST_NO_HOST_MEMORY: return ResultType::ErrorNotEnoughHostMemory;
	default: return ResultType::ErrorExternalLibrary;
	}
}

template <typename T>
ResultType compute_impl(NmfDescription<T>& d, ISummary* summary_iface) {
	if (g_context == nullptr) return ResultType::ErrorNotInitialized;

	if (d.initMethod == NmfInitializationMethod::CopyExisting && d.numRuns > 1) {
		log_summary("[WARNING] When using the CopyExisting initialization method, then no more than one run should be performed because of missing randomization!\n");
		d.numRuns = 1;
	}
	if (!d.useConstantBasisVectors && d.features > d.inputMatrix.columns) {
		log_error("[ERROR] Feature count has to be less than the matrix dimensions!");
		return ResultType::ErrorInvalidArgument;
	}

	nmfamd::AlgorithmParams prm;
	auto need = [&](const char* name, double& slot, const char* algo) -> bool {
		int idx = parameter_index(d.parameters, d.numParameters, name);
		if (idx < 0) {
			std::string msg = std::string("[ERROR] ") + algo + " algorithm requires parameter '" + name + "' to be set!";
			log_error(msg.c_str());
			return false;
		}
		slot = d.parameters[idx].value;
		return true;
	};
	switch (d.algorithm) {
	case NmfAlgorithm::Multiplicative: case NmfAlgorithm::ALS: break;
	case NmfAlgorithm::ACLS:
		if (!need("lambdaW", prm.lambdaW, "ACLS") || !need("lambdaH", prm.lambdaH, "ACLS")) return ResultType::ErrorInvalidArgument;
		break;
	case NmfAlgorithm::AHCLS:
		if (!need("lambdaW", prm.lambdaW, "AHCLS") || !need("lambdaH", prm.lambdaH, "AHCLS") ||
		    !need("alphaW", prm.alphaW, "AHCLS") || !need("alphaH", prm.alphaH, "AHCLS")) return ResultType::ErrorInvalidArgument;
		break;
	case NmfAlgorithm::GDCLS:
		if (!need("lambda", prm.lambda, "GDCLS")) return ResultType::ErrorInvalidArgument;
		break;
	case NmfAlgorithm::nsNMF:
		if (!need("theta", prm.theta, "nsNMF")) return ResultType::ErrorInvalidArgument;
		break;
	default:
		log_error("[ERROR] Chosen algorithm is not implemented!");
		return ResultType::ErrorInvalidArgument;
	}
	// extension switches ride on Parameter names the reference ignores (lookup is by name only,
	// Interface.cpp:41-49): absent => reference behaviour
	{
		// "nndsvd" (host_init.cpp): 0 / 1 / 2 only (3, -1 and NaN used to select the plain variant silently), and no more features than singular pairs exist
		int sidx = parameter_index(d.parameters, d.numParameters, "nndsvd");
		if (sidx >= 0) {
			const double v = d.parameters[sidx].value;
			if (!(v == 0.0 || v == 1.0 || v == 2.0)) {
				log_error("[ERROR] Parameter 'nndsvd' has to be 0 (NNDSVD), 1 (NNDSVDa) or 2 (NNDSVDar)!");
				return ResultType::ErrorInvalidArgument;
			}
			if (d.features > std::min(d.inputMatrix.rows, d.inputMatrix.columns)) {
				log_error("[ERROR] The NNDSVD start needs a feature count of at most min(rows, columns)!");
				return ResultType::ErrorInvalidArgument;
			}
		}
	}
	{
		int idx = parameter_index(d.parameters, d.numParameters, "divergence");
		if (idx >= 0) prm.divergence = d.parameters[idx].value;
		idx = parameter_index(d.parameters, d.numParameters, "sparseCompute");
		if (idx >= 0) prm.sparse_compute = d.parameters[idx].value;
		idx = parameter_index(d.parameters, d.numParameters, "precision");
		if (idx >= 0 && std::is_same<T, float>::value) prm.precision = d.parameters[idx].value;
		if ((prm.divergence != 0 || prm.sparse_compute != 0) && d.algorithm != NmfAlgorithm::Multiplicative) {
			log_error("[ERROR] 'divergence' / 'sparseCompute' are only available for the Multiplicative algorithm!");
			return ResultType::ErrorInvalidArgument;
		}
		if (prm.divergence != 0 && d.useConstantBasisVectors) {
			log_error("[ERROR] The KL-divergence update does not support constant basis vectors!");
			return ResultType::ErrorInvalidArgument;
		}
	}
	if (d.inputMatrix.rows == 0 || d.inputMatrix.columns == 0 || d.features == 0 ||
	    d.outputMatrixW.format != StorageFormat::Dense || d.outputMatrixH.format != StorageFormat::Dense) {
		log_error("[ERROR] Empty problem or non-dense output matrices!");
		return ResultType::ErrorInvalidArgument;
	}

	// Parameter "numGpus" = N > 1 (extension; the reference is single-GPU, SingleGpuDispatcher.h:36): column shards of V on N
	// rank threads inside this one call.  "shardMode": 0 reduce-scatter by row blocks of W, 1 replicated W update; absent: by the size of the m x r
	// exchange panel, as bench.py chooses -- 8 MB or more (config 4's 51 MB): row blocks; less (config 2's 2.6 MB): the replicated update, which at padded
	// rank 64 reads the ranks' panels in place and needs one rendezvous per iteration (sharded.cpp)
	int num_gpus = 1;
	int shard_mode = (double)sizeof(T) * (double)d.inputMatrix.rows * (double)nmfamd::padded_rank((int)d.features, sizeof(T)) >= 8e6 ? nmfamd::SHARD_ROW_BLOCKS : nmfamd::SHARD_REPLICATED;
	{
		int idx = parameter_index(d.parameters, d.numParameters, "numGpus");
		if (idx >= 0) num_gpus = (int)d.parameters[idx].value;
		idx = parameter_index(d.parameters, d.numParameters, "shardMode");
		if (idx >= 0) shard_mode = d.parameters[idx].value != 0 ? nmfamd::SHARD_REPLICATED : nmfamd::SHARD_ROW_BLOCKS;
		if (num_gpus > 1) {
			const bool mult = d.algorithm == NmfAlgorithm::Multiplicative || d.algorithm == NmfAlgorithm::nsNMF;
			if (num_gpus > 16 || (unsigned)num_gpus > d.inputMatrix.columns) {
				log_error("[ERROR] 'numGpus' > 1: at most 16 ranks and at least one column of the input matrix per rank!");
				return ResultType::ErrorInvalidArgument;
			}
			// GDCLS, the ALS family and the KL update: W is updated from the all-reduced sums on every rank (the row-block form covers the Frobenius multiplicative rule)
			if (!mult || prm.divergence != 0) shard_mode = nmfamd::SHARD_REPLICATED;
		}
	}

	if (!ensure_stream(*g_context)) {
		log_error("[ERROR] No usable HIP device: the factorisation kernels are gfx950 code objects and there is no CPU fallback!");
		return ResultType::ErrorExternalLibrary;
	}
	if (hipSetDevice(g_context->deviceID) != hipSuccess) { (void)hipGetLastError(); return ResultType::ErrorDeviceSelection; }

	SummaryImpl* summary = static_cast<SummaryImpl*>(summary_iface);
	if (summary) summary->reset();

	std::unique_ptr<runner::Runner<T>> run_ptr;
	if (num_gpus > 1) run_ptr.reset(new runner::TeamRunner<T>(d, prm, num_gpus, g_context->deviceID, shard_mode));
	else run_ptr.reset(new runner::SingleRunner<T>(d, prm, g_context->stream));
	runner::Runner<T>& engine = *run_ptr;
	nmfamd::Status st = engine.setup(d);
	if (st != nmfamd::ST_OK) {
		log_error("[ERROR] Device allocation or upload of the input matrix failed!");
		if (*engine.last_error()) log_error(engine.last_error());
		return from_status(st);
	}

	// what the loop reads is snapshotted here, like DispatcherConfig (source/nmf/Dispatcher.h:28-45)
	const unsigned numIterations = d.numIterations, numRuns = d.numRuns;
	const NmfThresholdType thresholdType = d.thresholdType;
	const double thresholdValue = d.thresholdValue;
	const UserInterruptCallback interrupt = d.callbackUserInterrupt;
	const bool constW = d.useConstantBasisVectors;
	// only W is initialised for the LS algorithms (their first step solves for H)
	const bool want_h = d.algorithm == NmfAlgorithm::Multiplicative || d.algorithm == NmfAlgorithm::nsNMF;

	// IAlgorithm's seed stream: constructed from the caller's seed, one draw per run
	std::mt19937 seed_stream(d.seed);

	if (allowed(Verbosity::Summary)) {
		char line[256];
		if (num_gpus > 1) std::snprintf(line, sizeof(line), " Executing %u run(s) of the '%s' algorithm on %d ranks from HIP device #%d (%s): \n", numRuns, algorithm_name(d.algorithm), num_gpus, g_context->deviceID, engine.describe());
		else std::snprintf(line, sizeof(line), " Executing %u run(s) of the '%s' algorithm on HIP device #%d: \n", numRuns, algorithm_name(d.algorithm), g_context->deviceID);
		log_summary(line);
	}

	bool interrupted = false;
	double best = std::numeric_limits<double>::max();
	for (unsigned run = 1; run <= numRuns; ++run) {
		if (run == 1) print_header(numRuns > 1);

		// initialize(): new seed into the caller's struct, then the factors
		const bool gdcls_const = d.algorithm == NmfAlgorithm::GDCLS && constW;  // GDCLS :147-157 skips the draw
		if (!gdcls_const) d.seed = static_cast<unsigned>(seed_stream());
		if (!gdcls_const) {
			if (d.initMethod == NmfInitializationMethod::CopyExisting &&
			    (d.outputMatrixW.format != StorageFormat::Dense || d.outputMatrixH.format != StorageFormat::Dense)) return ResultType::ErrorInvalidArgument;
			st = engine.init_run(d, want_h);
			if (st != nmfamd::ST_OK) { log_error("[ERROR] Initialisation of W / H failed!"); return from_status(st); }
		}
		if (constW) {
			st = engine.set_constant_w(d);
			if (st != nmfamd::ST_OK) return from_status(st);
		}
		engine.synchronize();

		auto started = std::chrono::high_resolution_clock::now();
		long long elapsed_ms = 0;
		double lastError = 0.0, delta = 0.0;
		unsigned iteration = 1;
		for (; iteration <= numIterations && !(interrupted = (interrupt != nullptr && interrupt())); ++iteration) {
			const bool computeError = iteration % 10 == 0 || iteration == numIterations;
			st = engine.iterate(computeError, constW);
			if (st != nmfamd::ST_OK) {
				log_error("[ERROR] A HIP call failed inside the iteration loop!");
				if (*engine.last_error()) log_error(engine.last_error());
				return from_status(st);
			}
			if (computeError) {
				// (the device would idle while the host waits for this iteration's error terms, sums them and decides: the first launch of the next
				//  iteration -- it writes scratch only -- goes out first; wasted when the threshold ends the run here)
				if (iteration < numIterations && !constW) {
					const nmfamd::Status ahead = engine.begin_next_iteration();
					if (ahead != nmfamd::ST_OK) {
						log_error("[ERROR] A HIP call failed while the next iteration's first launch was enqueued ahead!");
						if (*engine.last_error()) log_error(engine.last_error());
						return from_status(ahead);
					}
				}
				elapsed_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - started).count();
				const double current = thresholdType == NmfThresholdType::Frobenius ? engine.frobenius() : engine.rmsd();
				delta = current - lastError;
				print_row(numRuns > 1, false, run, numRuns, iteration, engine.frobenius(), engine.rmsd(), delta, elapsed_ms, "");
				if (lastError != 0.0 && std::fabs(delta) < thresholdValue) break;
				lastError = current;
			}
		}
		iteration = std::min(iteration, numIterations);

		if (interrupted) {
			print_row(numRuns > 1, true, run, numRuns, iteration, engine.frobenius(), engine.rmsd(), delta, elapsed_ms, "Aborted");
			break;
		}
		bool stored = false;
		if (engine.frobenius() < best) {
			if (summary) {
				ExecutionRecord rec = ExecutionRecord();
				rec.elapsedTime = elapsed_ms / 1000.0;
				rec.frobenius = engine.frobenius();
				rec.rmsd = engine.rmsd();
				rec.numIterations = iteration;
				summary->insert(rec);
			}
			st = engine.store(d);
			if (st != nmfamd::ST_OK) return from_status(st);
			best = engine.frobenius();
			stored = true;
		}
		print_row(numRuns > 1, true, run, numRuns, iteration, engine.frobenius(), engine.rmsd(), delta, elapsed_ms, stored ? "Stored" : "Discarded");
	}
	engine.synchronize();
	return interrupted ? ResultType::ErrorUserInterrupt : ResultType::Success;
}

} // namespace

// ---- exported C++ API ---------------------------------------------------------------------------

NMFGPU_EXPORT ResultType initialize() {
	if (g_context != nullptr) return ResultType::ErrorAlreadyInitialized;
	g_context = new DeviceContext();
	return ResultType::Success;
}

NMFGPU_EXPORT ResultType finalize() {
	if (g_context == nullptr) return ResultType::ErrorNotInitialized;
	drop_stream(*g_context);
	delete g_context;
	g_context = nullptr;
	return ResultType::Success;
}

NMFGPU_EXPORT int version() { return NMFGPU_VERSION; }

NMFGPU_EXPORT ResultType chooseGpu(unsigned index) {
	// the reference dereferences its context unchecked here (Interface.cpp:157); be kinder
	if (g_context == nullptr) return ResultType::ErrorNotInitialized;
	if (hipSetDevice(int(index)) != hipSuccess) { (void)hipGetLastError(); return ResultType::ErrorDeviceSelection; }
	drop_stream(*g_context);
	g_context->deviceID = int(index);
	return ResultType::Success;
}

NMFGPU_EXPORT unsigned getNumberOfGpu() {
	int num = 0;
	if (hipGetDeviceCount(&num) != hipSuccess) { (void)hipGetLastError(); return 0u; }
	return static_cast<unsigned>(num);
}

NMFGPU_EXPORT ResultType getInformationForGpuIndex(unsigned index, GpuInformation& info) {
	int old = 0;
	if (hipGetDevice(&old) != hipSuccess) { (void)hipGetLastError(); return ResultType::ErrorDeviceSelection; }
	if (hipSetDevice(int(index)) != hipSuccess) { (void)hipGetLastError(); return ResultType::ErrorDeviceSelection; }
	hipDeviceProp_t props;
	if (hipGetDeviceProperties(&props, int(index)) == hipSuccess) {
		// (the marketing name comes from libdrm's amdgpu.ids, which some installations lack: fall back to the architecture name)
		const char* name = props.name[0] != '\0' ? props.name : props.gcnArchName;
		std::strncpy(info.name, name, sizeof(info.name) - 1); info.name[sizeof(info.name) - 1] = '\0';
	}
	else std::strcpy(info.name, "N/A");
	hipError_t e = hipMemGetInfo(&info.freeMemory, &info.totalMemory);
	(void)hipSetDevice(old);
	if (e != hipSuccess) { info.freeMemory = 0; info.totalMemory = 0; return ResultType::ErrorExternalLibrary; }
	return ResultType::Success;
}

NMFGPU_EXPORT void setVerbosity(Verbosity verbosity) { g_verbosity.store(verbosity, std::memory_order_relaxed); }

NMFGPU_EXPORT ISummary* ISummary::create() { return new SummaryImpl(); }

NMFGPU_EXPORT ResultType compute(NmfDescription<float>& description, ISummary* summary) { return compute_impl(description, summary); }
NMFGPU_EXPORT ResultType compute(NmfDescription<double>& description, ISummary* summary) { return compute_impl(description, summary); }

NMFGPU_EXPORT ResultType computeKMeans(KMeansDescription<float>& desc, KMeansSummary* summary) {
	if (g_context == nullptr) return ResultType::ErrorNotInitialized;
	return hostinit::compute_kmeans<float>(desc, summary);
}
NMFGPU_EXPORT ResultType computeKMeans(KMeansDescription<double>& desc, KMeansSummary* summary) {
	if (g_context == nullptr) return ResultType::ErrorNotInitialized;
	return hostinit::compute_kmeans<double>(desc, summary);
}

} // namespace nmfgpu

// ---- exported C API (Interface.cpp:434-488) ---------------------------------------------------
extern "C" {

NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_initialize() { return nmfgpu::initialize(); }
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_finalize() { return nmfgpu::finalize(); }
NMFGPU_EXPORT int nmfgpu_version() { return nmfgpu::version(); }
NMFGPU_EXPORT void nmfgpu_set_verbosity(nmfgpu::Verbosity verbosity) { nmfgpu::setVerbosity(verbosity); }

NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_create_summary(nmfgpu::ISummary** summary) {
	if (summary == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	*summary = nmfgpu::ISummary::create();
	return nmfgpu::ResultType::Success;
}

NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_single(nmfgpu::NmfDescription<float>* description, nmfgpu::ISummary* summary) {
	if (description == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	return nmfgpu::compute(*description, summary);
}
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_double(nmfgpu::NmfDescription<double>* description, nmfgpu::ISummary* summary) {
	if (description == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	return nmfgpu::compute(*description, summary);
}
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_kmeans_single(nmfgpu::KMeansDescription<float>* desc) {
	if (desc == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	return nmfgpu::computeKMeans(*desc, nullptr);
}
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_kmeans_double(nmfgpu::KMeansDescription<double>* desc) {
	if (desc == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	return nmfgpu::computeKMeans(*desc, nullptr);
}
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_choose_gpu(unsigned index) { return nmfgpu::chooseGpu(index); }
NMFGPU_EXPORT unsigned nmfgpu_get_number_of_gpu() { return nmfgpu::getNumberOfGpu(); }
NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_get_information_for_gpu_index(unsigned index, nmfgpu::GpuInformation* info) {
	if (info == nullptr) return nmfgpu::ResultType::ErrorInvalidArgument;
	return nmfgpu::getInformationForGpuIndex(index, *info);
}

}
