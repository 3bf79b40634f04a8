// ref_harness.cpp -- C entry points around the two reference translation units that build
// in this image without the CUDA toolkit: source/nmf/Algorithm.cpp (the per-run seed stream)
// and source/nmf/Summary.cpp (best-run bookkeeping).  TEST INFRASTRUCTURE: built by
// oracle/Makefile into oracle/_ref/libnmfgpu_refhost.so from the sources where they lie under
// /root/reference (nothing is copied), used to pin oracle_seed_stream / oracle_summary_best_run
// and to generate tests/golden/ref_host_vectors.json (tests/golden/make_ref_host_vectors.py).
//
// This file is our own glue: it only derives from / calls the reference classes.
#include <nmf/Algorithm.h>
#include <nmf/Summary.h>
#include <cstdint>

namespace {
// IAlgorithm's constructor and generateRandomNumber() are protected; a do-nothing subclass
// exposes them without touching any device code.
class SeedProbe : public nmfgpu::IAlgorithm {
public:
	explicit SeedProbe(unsigned seed) : nmfgpu::IAlgorithm(seed) { }
	unsigned next() { return generateRandomNumber(); }
	void computeIteration(bool) override { }
	nmfgpu::ResultType allocateMemory() override { return nmfgpu::ResultType::Success; }
	void deallocateMemory() override { }
	void initialize() override { }
	void storeFactorization() override { }
	const char* name() const override { return "seed probe"; }
};
}

extern "C" {

void ref_seed_stream(uint32_t seed, int count, uint32_t* out) {
	SeedProbe probe(seed);
	for (int i = 0; i < count; ++i) out[i] = probe.next();
}

// Feeds `count` records with the given frobenius values through Summary::reset/insert and
// reports bestRun() and recordCount(); also reads record(index) back into `readback`.
unsigned ref_summary_best_run(const double* frobenius, int count, unsigned* recordCount, double* readback) {
	nmfgpu::Summary* s = new nmfgpu::Summary();
	s->reset();
	for (int i = 0; i < count; ++i) {
		nmfgpu::ExecutionRecord rec = nmfgpu::ExecutionRecord();
		rec.frobenius = frobenius[i];
		rec.numIterations = unsigned(i + 1);
		s->insert(rec);
	}
	unsigned best = count > 0 ? s->bestRun() : 0u;
	if (recordCount) *recordCount = s->recordCount();
	if (readback)
		for (int i = 0; i < count; ++i) {
			nmfgpu::ExecutionRecord rec;
			s->record(unsigned(i), rec);
			readback[i] = rec.frobenius;
		}
	s->destroy();
	return best;
}

}
