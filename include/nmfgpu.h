// nmfgpu.h -- public boundary of the MI355X-native NMF engine.
//
// This header declares, name for name and byte for byte, the application
// binary interface of nmfgpu v0.2.3 so that existing callers (the upstream
// example program, the nmfgpu4R binding) compile and link against this
// library unchanged.  It is an independent restatement of that interface:
// every declaration cites the upstream declaration it replaces as
// `ref: include/nmfgpu.h:<line>`.  The layout facts it must honour
// (checked by the static_asserts at the end of this file and by
// tests/test_abi.py):
//
//   * everything between the two pack pragmas is laid out under
//     `#pragma pack(4)`                                  (ref: :47, :351)
//   * every enum is an `enum class` with underlying type int, numbered from 0
//     in declaration order                               (ref: :52-134, :177-186)
//   * ISummary's virtual table is: destroy, bestRun, record, recordCount,
//     then the (protected) virtual destructor            (ref: :149-175)
//   * C++ entry points live in namespace nmfgpu with Itanium mangling, and a
//     second set of unmangled `nmfgpu_*` C entry points wraps them
//                                                        (ref: :276-299, :329-349)
//
// Nothing behind this boundary is shared with the upstream implementation.
// MI355X-specific extensions never change these declarations: they are
// selected through additional `Parameter` names (ignored by upstream) and
// through the separate header nmfgpu_amd.h.
#pragma once

#include <cstddef>

// ref: include/nmfgpu.h:28-31 -- version triple and its packed form.
#define NMFGPU_MAJOR 0
#define NMFGPU_MINOR 2
#define NMFGPU_PATCH 3
#define NMFGPU_VERSION ((NMFGPU_MAJOR << 24) | (NMFGPU_MINOR << 16) | NMFGPU_PATCH)

// ref: include/nmfgpu.h:33-45 -- symbol visibility.  Callers define
// NMFGPU_STATIC_LINKING (the upstream example does) or nothing at all; on
// ELF platforms the macro is empty for users and "default visibility" while
// this library itself is being built.
#if defined(_WIN32) && !defined(NMFGPU_STATIC_LINKING)
#  if defined(NMFGPU_EXPORTING)
#    define NMFGPU_EXPORT __declspec(dllexport)
#  else
#    define NMFGPU_EXPORT __declspec(dllimport)
#  endif
#elif defined(NMFGPU_EXPORTING)
#  define NMFGPU_EXPORT __attribute__((visibility("default")))
#else
#  define NMFGPU_EXPORT
#endif

#pragma pack(push, 4)

namespace nmfgpu {

// ---------------------------------------------------------------------------
// Enumerations
// ---------------------------------------------------------------------------

// ref: include/nmfgpu.h:52-77.  Status code returned by every entry point.
enum class ResultType {
	Success = 0,                 // call completed
	ErrorAlreadyInitialized,     // initialize() twice on one thread
	ErrorNotInitialized,         // compute()/finalize() before initialize()
	ErrorInvalidArgument,        // missing Parameter, features > columns, null out-pointer ...
	ErrorNotEnoughHostMemory,
	ErrorNotEnoughDeviceMemory,
	ErrorExternalLibrary,        // here: a HIP runtime call failed
	ErrorUserInterrupt,          // the interrupt callback returned true
	ErrorDeviceSelection,        // chooseGpu / getInformationForGpuIndex on a bad index
};

// ref: include/nmfgpu.h:80-100.  How W and H get their starting values.
enum class NmfInitializationMethod {
	CopyExisting,             // W, H are read from outputMatrixW / outputMatrixH
	AllRandomValues,          // uniform (0,1], W and H drawn from the same per-run seed
	MeanColumns,              // W(:,k) = mean of five random columns of V, H random
	KMeansAndRandomValues,    // W = k-means centroids, H random
	KMeansAndAbsoluteWTV,     // declared upstream, never handled there
	KMeansAndNonNegativeWTV,  // W = k-means centroids, H = max(0, W^T V)
	EInNMF,                   // W = k-means centroids, H from fuzzy memberships
};

// ref: include/nmfgpu.h:102-105.  Which error measure drives the stop test.
enum class NmfThresholdType {
	Frobenius,
	RMSD
};

// ref: include/nmfgpu.h:107-114.  Factorisation algorithm.
enum class NmfAlgorithm {
	Multiplicative,  // Lee-Seung multiplicative update, Frobenius objective
	GDCLS,           // least-squares H (lambda), multiplicative W
	ALS,             // alternating least squares
	ACLS,            // ALS with ridge terms lambdaW / lambdaH
	AHCLS,           // ACLS with Hoyer sparseness terms alphaW / alphaH
	nsNMF,           // non-smooth NMF (theta)
};

// ref: include/nmfgpu.h:117-126.  Console output level (process wide).
enum class Verbosity {
	None,         // errors only
	Summary,      // the progress table (default)
	Informative,  // convergence details
	Debugging     // everything
};

// ref: include/nmfgpu.h:129-134.  Index origin of sparse index arrays.
enum class IndexBase {
	Zero,
	One
};

// ref: include/nmfgpu.h:177-186.  Storage scheme of a MatrixDescription.
enum class StorageFormat {
	Dense,  // column-major with leading dimension
	CSR,    // compressed sparse row
	CSC,    // compressed sparse column
	COO     // coordinate triplets
};

// ---------------------------------------------------------------------------
// Plain-data descriptors (caller owned; all pointers are HOST pointers)
// ---------------------------------------------------------------------------

// ref: include/nmfgpu.h:136.  Polled once per iteration on the calling
// thread; returning true aborts the factorisation.
typedef bool(*UserInterruptCallback)();

// ref: include/nmfgpu.h:138-147.  One stored run.  sparsityW / sparsityH are
// declared upstream but never written (source/nmf/SingleGpuDispatcher.cpp:217-222).
struct ExecutionStatistic {
	double frobenius;        // ||V - W H||_F as defined by the trace formula
	double rmsd;             // frobenius / sqrt(rows * columns)
	double elapsedTime;      // seconds, wall clock of the run up to its last error check
	double sparsityW;
	double sparsityH;
	unsigned numIterations;  // iterations actually executed
};
typedef ExecutionStatistic ExecutionRecord;

// ref: include/nmfgpu.h:149-175.  Caller-owned collection of the runs that
// improved on the best error so far.  Obtain with create() (or
// nmfgpu_create_summary), release with destroy().  The order of the virtual
// members is part of the ABI.
class ISummary {
public:
	NMFGPU_EXPORT static ISummary* create();

	virtual void destroy() = 0;
	virtual unsigned bestRun() const = 0;
	virtual void record(unsigned index, ExecutionRecord& record) const = 0;
	virtual unsigned recordCount() const = 0;

protected:
	virtual ~ISummary() { }
};

// ref: include/nmfgpu.h:188-232.  A host matrix in one of four storage
// schemes.  The four anonymous structs overlay each other; sparse index
// arrays are 32-bit and honour `base`.
template<typename NumericType>
struct MatrixDescription {
	unsigned rows;
	unsigned columns;
	StorageFormat format;
	union {
		struct {
			NumericType* values;        // column-major, element (i,j) at values[i + j*leadingDimension]
			unsigned leadingDimension;
		} dense;

		struct {
			NumericType* values;
			int* rowPtr;                // rows + 1 entries
			int* columnIndices;         // nnz entries
			unsigned nnz;
			IndexBase base;
		} csr;

		struct {
			NumericType* values;
			int* columnPtr;             // columns + 1 entries
			int* rowIndices;            // nnz entries
			unsigned nnz;
			IndexBase base;
		} csc;

		struct {
			NumericType* values;
			int* rowIndices;            // nnz entries, sorted by row (an uncompressed CSR)
			int* columnIndices;         // nnz entries
			unsigned nnz;
			IndexBase base;
		} coo;
	};
};

// ref: include/nmfgpu.h:234-237.  Named algorithm parameter; lookup is by
// strcmp on `name` (source/common/Interface.cpp:41-49), unknown names are ignored.
struct Parameter {
	const char* name;
	double value;
};

// ref: include/nmfgpu.h:239-273.  Complete description of one factorisation
// job.  The library writes `seed` (every run) and `numRuns` (clamped to 1 for
// CopyExisting) back into this struct, and writes W and H of every run that
// improves on the best error into outputMatrixW / outputMatrixH.
template<typename NumericType>
struct NmfDescription {
	NmfAlgorithm algorithm;
	bool useConstantBasisVectors;                  // keep W = outputMatrixW fixed, fit H only
	MatrixDescription<NumericType> inputMatrix;    // V: attributes in rows, samples in columns
	int* inputLabels;                              // optional, unused by every algorithm
	MatrixDescription<NumericType> outputMatrixW;  // dense, rows x features
	MatrixDescription<NumericType> outputMatrixH;  // dense, features x columns
	unsigned features;                             // r
	NmfInitializationMethod initMethod;
	unsigned numIterations;
	unsigned numRuns;
	unsigned seed;
	NmfThresholdType thresholdType;
	double thresholdValue;
	UserInterruptCallback callbackUserInterrupt;   // may be null
	Parameter* parameters;
	unsigned numParameters;
};

// ref: include/nmfgpu.h:288-292.
struct GpuInformation {
	char name[256];
	size_t totalMemory;
	size_t freeMemory;
};

// ref: include/nmfgpu.h:301-310.  k-means job (samples are the columns).
template<typename NumericType>
struct KMeansDescription {
	MatrixDescription<NumericType> inputMatrix;
	MatrixDescription<NumericType> outputMatrixClusters;  // dense, rows x numClusters
	unsigned* outputMemberships;                          // optional, one entry per column
	unsigned numClusters;
	unsigned numIterations;
	unsigned seed;
	double thresholdValue;
};

// ref: include/nmfgpu.h:312-324.  Declared upstream, never filled there
// (source/common/Interface.cpp:404-406).
struct KMeansSummary {
	unsigned iterations;
	double betweenSS;
	double* withinSS;
	double totalWithinSS;
	double totalSS;
};

// ---------------------------------------------------------------------------
// C++ entry points (ref: include/nmfgpu.h:276-299, 326-327)
// ---------------------------------------------------------------------------

// Per-thread bring-up / tear-down of the device context.
NMFGPU_EXPORT ResultType initialize();
NMFGPU_EXPORT ResultType finalize();

// NMFGPU_VERSION of the library that was linked.
NMFGPU_EXPORT int version();

// Device selection and introspection.
NMFGPU_EXPORT ResultType chooseGpu(unsigned index);
NMFGPU_EXPORT unsigned getNumberOfGpu();
NMFGPU_EXPORT ResultType getInformationForGpuIndex(unsigned index, GpuInformation& info);

NMFGPU_EXPORT void setVerbosity(Verbosity verbosity);

// The factorisation itself; `summary` may be null.
NMFGPU_EXPORT ResultType compute(NmfDescription<float>& description, ISummary* summary);
NMFGPU_EXPORT ResultType compute(NmfDescription<double>& description, ISummary* summary);

NMFGPU_EXPORT ResultType computeKMeans(KMeansDescription<float>& desc, KMeansSummary* summary);
NMFGPU_EXPORT ResultType computeKMeans(KMeansDescription<double>& desc, KMeansSummary* summary);

} // namespace nmfgpu

// ---------------------------------------------------------------------------
// C entry points (ref: include/nmfgpu.h:329-349) -- what a dlopen/FFI caller binds.
// ---------------------------------------------------------------------------
extern "C" {
	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_initialize();
	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_finalize();
	NMFGPU_EXPORT int nmfgpu_version();
	NMFGPU_EXPORT void nmfgpu_set_verbosity(nmfgpu::Verbosity verbosity);

	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_create_summary(nmfgpu::ISummary** summary);

	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_single(nmfgpu::NmfDescription<float>* description, nmfgpu::ISummary* summary);
	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_double(nmfgpu::NmfDescription<double>* description, nmfgpu::ISummary* summary);

	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_kmeans_single(nmfgpu::KMeansDescription<float>* desc);
	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_compute_kmeans_double(nmfgpu::KMeansDescription<double>* desc);

	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_choose_gpu(unsigned index);
	NMFGPU_EXPORT unsigned nmfgpu_get_number_of_gpu();
	NMFGPU_EXPORT nmfgpu::ResultType nmfgpu_get_information_for_gpu_index(unsigned index, nmfgpu::GpuInformation* info);
}

#pragma pack(pop)

// ---------------------------------------------------------------------------
// Layout guard (LP64 only).  The numbers are the upstream header's layout as
// measured with offsetof/sizeof on x86-64; a mismatch
// here means this header stopped being a drop-in.
// ---------------------------------------------------------------------------
#if defined(__LP64__) && !defined(NMFGPU_NO_LAYOUT_GUARD)
static_assert(sizeof(nmfgpu::MatrixDescription<float>) == 44 && sizeof(nmfgpu::MatrixDescription<double>) == 44, "MatrixDescription layout");
static_assert(sizeof(nmfgpu::NmfDescription<float>) == 200 && sizeof(nmfgpu::NmfDescription<double>) == 200, "NmfDescription layout");
static_assert(sizeof(nmfgpu::KMeansDescription<float>) == 116, "KMeansDescription layout");
static_assert(sizeof(nmfgpu::ExecutionStatistic) == 44, "ExecutionStatistic layout");
static_assert(sizeof(nmfgpu::Parameter) == 16, "Parameter layout");
static_assert(sizeof(nmfgpu::GpuInformation) == 272, "GpuInformation layout");
static_assert(sizeof(nmfgpu::KMeansSummary) == 36, "KMeansSummary layout");
static_assert(sizeof(nmfgpu::ResultType) == 4 && sizeof(nmfgpu::StorageFormat) == 4, "enum width");
#endif
