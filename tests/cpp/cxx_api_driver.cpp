// cxx_api_driver.cpp -- test program: a C++ caller of the drop-in boundary, written against include/nmfgpu.h only
// (the way the reference's example program and the R binding's shim use the library: namespace nmfgpu, mangled
// entry points, the ISummary vtable).  Reads a problem from a binary file, runs nmfgpu::compute in double with
// CopyExisting initialisation, writes W, H and the summary record back.  Driven by tests/test_gpu_cxx_api.py.
//
// file layout (little endian): int32 m, n, r, algorithm, iterations; then V (m*n), W (m*r), H (r*n) as float64, column-major
#include <nmfgpu.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
	if (argc < 3) { std::fprintf(stderr, "usage: cxx_api_driver <in> <out>\n"); return 2; }
	std::FILE* f = std::fopen(argv[1], "rb");
	if (!f) return 3;
	int32_t hdr[5];
	if (std::fread(hdr, sizeof(int32_t), 5, f) != 5) return 3;
	const unsigned m = hdr[0], n = hdr[1], r = hdr[2];
	std::vector<double> V((size_t)m * n), W((size_t)m * r), H((size_t)r * n);
	if (std::fread(V.data(), 8, V.size(), f) != V.size() || std::fread(W.data(), 8, W.size(), f) != W.size() ||
	    std::fread(H.data(), 8, H.size(), f) != H.size()) return 3;
	std::fclose(f);

	if (nmfgpu::initialize() != nmfgpu::ResultType::Success) return 4;
	nmfgpu::setVerbosity(nmfgpu::Verbosity::None);
	if (nmfgpu::getNumberOfGpu() < 1) return 5;

	nmfgpu::NmfDescription<double> d;
	d.algorithm = static_cast<nmfgpu::NmfAlgorithm>(hdr[3]);
	d.useConstantBasisVectors = false;
	d.inputMatrix.rows = m; d.inputMatrix.columns = n; d.inputMatrix.format = nmfgpu::StorageFormat::Dense;
	d.inputMatrix.dense.values = V.data(); d.inputMatrix.dense.leadingDimension = m;
	d.inputLabels = nullptr;
	d.outputMatrixW.rows = m; d.outputMatrixW.columns = r; d.outputMatrixW.format = nmfgpu::StorageFormat::Dense;
	d.outputMatrixW.dense.values = W.data(); d.outputMatrixW.dense.leadingDimension = m;
	d.outputMatrixH.rows = r; d.outputMatrixH.columns = n; d.outputMatrixH.format = nmfgpu::StorageFormat::Dense;
	d.outputMatrixH.dense.values = H.data(); d.outputMatrixH.dense.leadingDimension = r;
	d.features = r;
	d.initMethod = nmfgpu::NmfInitializationMethod::CopyExisting;
	d.numIterations = (unsigned)hdr[4];
	d.numRuns = 1;
	d.seed = 1234;
	d.thresholdType = nmfgpu::NmfThresholdType::Frobenius;
	d.thresholdValue = 0.0;
	d.callbackUserInterrupt = nullptr;
	nmfgpu::Parameter parameters[] = {{"lambdaH", 0.01}, {"lambdaW", 0.01}, {"alphaH", 0.01}, {"alphaW", 0.01}, {"lambda", 0.01}, {"theta", 0.5}};
	d.parameters = parameters;
	d.numParameters = 6;

	nmfgpu::ISummary* summary = nmfgpu::ISummary::create();
	const auto t0 = std::chrono::steady_clock::now();
	const nmfgpu::ResultType res = nmfgpu::compute(d, summary);
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (res != nmfgpu::ResultType::Success) { std::fprintf(stderr, "compute failed: %d\n", (int)res); return 6; }
	if (summary->recordCount() != 1 || summary->bestRun() != 0) return 7;
	nmfgpu::ExecutionRecord rec;
	summary->record(0, rec);
	summary->destroy();
	nmfgpu::finalize();

	std::FILE* o = std::fopen(argv[2], "wb");
	if (!o) return 8;
	double meta[4] = {rec.frobenius, rec.rmsd, (double)rec.numIterations, seconds};
	std::fwrite(meta, 8, 4, o);
	std::fwrite(W.data(), 8, W.size(), o);
	std::fwrite(H.data(), 8, H.size(), o);
	std::fclose(o);
	std::printf("ok %u x %u r=%u alg=%d iterations=%u frobenius=%.9f seconds=%.4f\n", m, n, r, hdr[3], rec.numIterations, rec.frobenius, seconds);
	return 0;
}
