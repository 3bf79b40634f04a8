// host_init.h -- host-side initialisation strategies and k-means (internal header).
#pragma once

#include "../../include/nmfgpu.h"

namespace nmfgpu {
namespace hostinit {

// W (m x r, ld m) and optionally H (r x n, ld r) for MeanColumns / KMeans* / EInNMF.
// Returns false for methods it does not handle.
template <typename T>
bool initialize(const NmfDescription<T>& d, T* W, T* H);

template <typename T>
ResultType compute_kmeans(KMeansDescription<T>& desc, KMeansSummary* summary);

// Parameter{"nndsvd", 0 | 1 | 2} (NNDSVD / NNDSVDa / NNDSVDar; Boutsidis & Gallopoulos 2008) overrides initMethod: -1 when absent.  initialize() then fills W and H
// from the truncated SVD of V, computed on the host.
int nndsvd_variant(const Parameter* parameters, unsigned count);

// Same counter-based uniform (0, 1] generator as the device fill (kernels.hip, k_fill_uniform).
double uniform01(unsigned long long seed, unsigned long long index, bool single_precision);

} // namespace hostinit
} // namespace nmfgpu
