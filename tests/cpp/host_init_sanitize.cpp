// Sanitizer driver for the host-side initialisers (nmfgpu_amd/csrc/host_init.cpp): compiled by tests/test_host_init.py with
// -fsanitize=address,undefined against the host translation unit alone (no HIP), run on odd shapes: k-means (dense float / double),
// every initialisation method host_init serves.  Exit code 0 and an empty sanitizer report are the test.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../nmfgpu_amd/csrc/host_init.h"

using namespace nmfgpu;

template <typename T>
static int run(unsigned m, unsigned n, unsigned k) {
	std::vector<T> V((size_t)m * n), W((size_t)m * k), H((size_t)k * n), C((size_t)m * k);
	std::vector<unsigned> memb(n);
	unsigned long long s = 12345u + m * 7 + n;
	for (auto& v : V) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (T)((s >> 40) % 1000) / (T)250 + (T)0.01; }
	KMeansDescription<T> kd = {};
	kd.inputMatrix.rows = m; kd.inputMatrix.columns = n; kd.inputMatrix.format = StorageFormat::Dense;
	kd.inputMatrix.dense.values = V.data(); kd.inputMatrix.dense.leadingDimension = m;
	kd.outputMatrixClusters.rows = m; kd.outputMatrixClusters.columns = k; kd.outputMatrixClusters.format = StorageFormat::Dense;
	kd.outputMatrixClusters.dense.values = C.data(); kd.outputMatrixClusters.dense.leadingDimension = m;
	kd.outputMemberships = memb.data();
	kd.numClusters = k; kd.numIterations = 7; kd.seed = 3; kd.thresholdValue = 0.0;
	if (hostinit::compute_kmeans(kd, nullptr) != ResultType::Success) return 1;
	const NmfInitializationMethod methods[] = {NmfInitializationMethod::MeanColumns, NmfInitializationMethod::KMeansAndRandomValues,
	                                           NmfInitializationMethod::KMeansAndNonNegativeWTV, NmfInitializationMethod::KMeansAndAbsoluteWTV,
	                                           NmfInitializationMethod::EInNMF};
	for (NmfInitializationMethod im : methods) {
		NmfDescription<T> d = {};
		d.algorithm = NmfAlgorithm::Multiplicative;
		d.inputMatrix = kd.inputMatrix;
		d.outputMatrixW.rows = m; d.outputMatrixW.columns = k; d.outputMatrixW.format = StorageFormat::Dense;
		d.outputMatrixW.dense.values = W.data(); d.outputMatrixW.dense.leadingDimension = m;
		d.outputMatrixH.rows = k; d.outputMatrixH.columns = n; d.outputMatrixH.format = StorageFormat::Dense;
		d.outputMatrixH.dense.values = H.data(); d.outputMatrixH.dense.leadingDimension = k;
		d.features = k; d.initMethod = im; d.numIterations = 1; d.numRuns = 1; d.seed = 11;
		if (!hostinit::initialize(d, W.data(), H.data())) return 2;
		for (T v : W) if (!(v == v)) return 3;          // no NaN
		for (T v : H) if (!(v == v)) return 4;
	}
	return 0;
}

int main() {
	const unsigned shapes[][3] = {{1, 2, 1}, {3, 5, 2}, {33, 65, 7}, {64, 32, 31}, {130, 257, 9}};      // (k < n: the reference rejects the rest)
	for (auto& sh : shapes) {
		if (int rc = run<float>(sh[0], sh[1], sh[2])) { std::printf("float %u x %u k=%u: %d\n", sh[0], sh[1], sh[2], rc); return rc; }
		if (int rc = run<double>(sh[0], sh[1], sh[2])) { std::printf("double %u x %u k=%u: %d\n", sh[0], sh[1], sh[2], rc); return rc; }
	}
	std::printf("ok\n");
	return 0;
}
