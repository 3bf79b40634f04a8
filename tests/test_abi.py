"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol the two
headers declare, keeps the reference's struct layout, and refuses to compute without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import nmfgpu_amd as na

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CXX_SYMBOLS = [
    "_ZN6nmfgpu10initializeEv", "_ZN6nmfgpu8finalizeEv", "_ZN6nmfgpu7versionEv", "_ZN6nmfgpu9chooseGpuEj",
    "_ZN6nmfgpu14getNumberOfGpuEv", "_ZN6nmfgpu25getInformationForGpuIndexEjRNS_14GpuInformationE",
    "_ZN6nmfgpu12setVerbosityENS_9VerbosityE",
    "_ZN6nmfgpu7computeERNS_14NmfDescriptionIfEEPNS_8ISummaryE", "_ZN6nmfgpu7computeERNS_14NmfDescriptionIdEEPNS_8ISummaryE",
    "_ZN6nmfgpu13computeKMeansERNS_17KMeansDescriptionIfEEPNS_13KMeansSummaryE",
    "_ZN6nmfgpu13computeKMeansERNS_17KMeansDescriptionIdEEPNS_13KMeansSummaryE",
    "_ZN6nmfgpu8ISummary6createEv",
]


def _declared_c_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(nmfgpu_[a-z_0-9]+|nmfamd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = na.library()
    names = _declared_c_symbols("nmfgpu.h") + _declared_c_symbols("nmfgpu_amd.h")
    assert len(names) >= 12 + 25
    for name in names + CXX_SYMBOLS:
        assert hasattr(lib, name), f"{name} is declared but not exported"


def test_struct_layout_matches_reference_table():
    # SURVEY.md section 8b, measured on the reference header with offsetof
    assert C.sizeof(na.MatrixDescription) == 44
    assert C.sizeof(na.NmfDescription) == 200
    assert C.sizeof(na.KMeansDescription) == 116
    assert C.sizeof(na.ExecutionRecord) == 44
    assert C.sizeof(na.Parameter) == 16
    assert C.sizeof(na.GpuInformation) == 272
    d = na.NmfDescription
    offs = {f: getattr(d, f).offset for f, _ in d._fields_}
    assert offs == {"algorithm": 0, "useConstantBasisVectors": 4, "inputMatrix": 8, "inputLabels": 52, "outputMatrixW": 60,
                    "outputMatrixH": 104, "features": 148, "initMethod": 152, "numIterations": 156, "numRuns": 160, "seed": 164,
                    "thresholdType": 168, "thresholdValue": 172, "callbackUserInterrupt": 180, "parameters": 188, "numParameters": 196}


def test_header_is_layout_compatible_with_reference_header():
    """Compile a probe against OUR header and (when present) the reference header: same offsets."""
    probe = r'''
#include <nmfgpu.h>
#include <cstdio>
#include <cstddef>
int main() {
  using namespace nmfgpu;
  printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(MatrixDescription<float>), sizeof(NmfDescription<double>), sizeof(KMeansDescription<float>),
         sizeof(ExecutionStatistic), sizeof(Parameter), sizeof(GpuInformation), sizeof(KMeansSummary));
  printf("%zu %zu %zu %zu %zu ", offsetof(NmfDescription<float>, thresholdValue), offsetof(NmfDescription<float>, parameters),
         offsetof(MatrixDescription<double>, csr.nnz), offsetof(MatrixDescription<double>, csr.base), offsetof(KMeansDescription<double>, thresholdValue));
  printf("%d %d %d %d\n", (int)ResultType::ErrorDeviceSelection, (int)NmfInitializationMethod::EInNMF, (int)NmfAlgorithm::nsNMF, (int)StorageFormat::COO);
  return NMFGPU_VERSION == 0x00020003 ? 0 : 1;
}
'''
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.cpp")
        open(src, "w").write(probe)
        for inc in (os.path.join(ROOT, "include"), "/root/reference/include"):
            if not os.path.isdir(inc):
                continue
            exe = os.path.join(td, "probe")
            subprocess.check_call(["g++", "-std=c++11", "-Wno-invalid-offsetof", "-I", inc, src, "-o", exe])
            outs.append(subprocess.check_output([exe]).decode())
    assert outs[0].split() == "44 200 116 44 16 272 36 172 188 36 40 108 8 6 5 3".split()
    assert all(o == outs[0] for o in outs)


def test_reference_example_compiles_and_links_unmodified():
    example = "/root/reference/example/main.cpp"
    if not os.path.exists(example):
        pytest.skip("reference tree not on this box")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "nmfgpu_example")
        libdir = os.path.dirname(na.library_path())
        subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), example, "-o", exe,
                               "-L", libdir, "-lnmfgpu64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
        assert os.path.exists(exe)


def test_initialize_semantics_without_gpu():
    assert na.version() == 0x00020003
    assert na.finalize() == na.ResultType.ErrorNotInitialized
    assert na.initialize() == na.ResultType.Success
    assert na.initialize() == na.ResultType.ErrorAlreadyInitialized
    assert na.finalize() == na.ResultType.Success


def test_compute_fails_loudly_without_a_device():
    if na.device_count() > 0:
        pytest.skip("a GPU is present")
    rng = np.random.default_rng(0)
    V = np.asfortranarray(rng.random((20, 10)).astype(np.float32))
    W = np.asfortranarray(rng.random((20, 3)).astype(np.float32)); H = np.asfortranarray(rng.random((3, 10)).astype(np.float32))
    W0 = W.copy()
    assert na.compute(V, W, H, iterations=5) == na.ResultType.ErrorNotInitialized
    assert na.initialize() == na.ResultType.Success
    try:
        na.set_verbosity(na.Verbosity.Nothing)
        assert na.compute(V, W, H, iterations=5) == na.ResultType.ErrorExternalLibrary   # no CPU fallback
        assert np.array_equal(W, W0)
        with pytest.raises(na.EngineError):
            na.Engine(20, 10, 3)
        with pytest.raises(na.EngineError):
            na.op_factor_product(V, np.asfortranarray(rng.random((3, 10)).astype(np.float32)))
    finally:
        na.finalize()


def test_summary_vtable_through_c_layer():
    s = na.Summary()
    assert s.record_count() == 0 and s.best_run() == 0
    s.destroy()


def test_package_does_not_import_the_oracle():
    import sys
    mods = [m for m in sys.modules if m.startswith("oracle")]
    code = "import sys, nmfgpu_amd; print([m for m in sys.modules if m.startswith('oracle')])"
    out = subprocess.check_output(["python", "-c", code], cwd=ROOT).decode().strip()
    assert out == "[]", out
    # nothing under the product package may import, link, load or execute anything under oracle/
    uses = re.compile(r"import\s+oracle|from\s+oracle|from\s+\.+oracle|nmf_oracle|libnmf_oracle|oracle[/\\.]py|oracle/|dlopen")
    for root, _, files in os.walk(os.path.join(ROOT, "nmfgpu_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                if f == "comm.hip":
                    # the one run-time load in the product: RCCL's shared library (multi-GPU runs only); every library name
                    # in that file must be RCCL's
                    names = re.findall(r'"([^"]*\.so[^"]*)"', text)
                    assert names and all("rccl" in n for n in names), names
                    text = text.replace("dlopen", "")
                assert not uses.search(text), f


def test_no_packed_fp32_instruction_reads_a_scalar_register():
    """gfx950 rule found in round 3 (docs/DESIGN_r05.md section 11): v_pk_fma / mul / add_f32 with a scalar-register source returned wrong low
    halves in lanes 48..63 whenever another kernel's waves shared the SIMD.  The build refuses such code (nmfgpu_amd/build.py);
    this re-checks the objects the loaded library was linked from, and that the check itself still sees the offending form."""
    from nmfgpu_amd import build as b
    objs = [os.path.join(b.OBJDIR, os.path.splitext(s)[0] + ".o") for s in b.SOURCES if s.endswith(".hip")]
    assert len(objs) >= 10 and all(os.path.exists(o) for o in objs)
    for o in objs:
        assert b.packed_scalar_sources(o) == [], o
    assert "-fno-slp-vectorize" in b.DEVICE_FLAGS
    assert b._PACKED_SCALAR.search("v_pk_fma_f32 v[14:15], s[38:39], v[14:15], v[36:37] op_sel:[0,0,1] op_sel_hi:[0,1,1]")
    assert b._PACKED_SCALAR.search("v_pk_add_f32 v[18:19], s[30:31], v[18:19] op_sel:[1,0]")
    assert b._PACKED_SCALAR.search("v_pk_mul_f32 v[2:3], v[2:3], s4")
    assert not b._PACKED_SCALAR.search("v_pk_fma_f32 v[22:23], v[36:37], v[22:23], v[42:43] op_sel_hi:[0,1,0]")
    assert not b._PACKED_SCALAR.search("v_pk_mul_f32 v[2:3], v[2:3], v[36:37]")


def test_shipped_library_reads_only_the_documented_environment_variables():
    """Product-library hygiene: the A/B switches, forced kernel forms and rehearsal modes live behind tuning_env() and exist in the measurement build
    only (csrc/tuning.h); the shipped libnmfgpu64.so may name nothing but the behavioural variables listed there.  NMFAMD_SHARD_REHEARSE in particular -- a value > 1
    makes a rank update 1 / N of W's rows -- must not be readable from the product."""
    import re
    import subprocess
    from nmfgpu_amd import _lib
    allowed = {"NMFAMD_COMM", "NMFAMD_SELFTEST", "NMFAMD_HOST_THREADS", "NMFAMD_MALL_MB", "NMFAMD_KL_BLOCK_KB", "NMFAMD_ONE_IMAGE",
               "NMFAMD_FORCE_VALU", "NMFAMD_NO_FUSED_MU", "NMFAMD_GRAM_PARTIALS", "NMFAMD_FP_TILE", "NMFAMD_SPARSE_SETUP"}
    path = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib", "libnmfgpu64.so")
    text = subprocess.run(["strings", "-n", "8", path], check=True, capture_output=True, text=True).stdout
    found = set(re.findall(r"NMFAMD_[A-Z0-9_]+", text))
    assert found <= allowed, sorted(found - allowed)
    header = open(os.path.join(ROOT, "nmfgpu_amd", "csrc", "tuning.h")).read()
    for name in sorted(found):
        assert name in header, f"{name} is read by the shipped library but not documented in csrc/tuning.h"
