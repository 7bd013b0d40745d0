import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # The GPU box shows 256 CPUs under a cgroup quota of 16: left alone torch starts 128 intra-op threads and the oracle's
    # fp64 pass of the B7 2 x 64 x 64 case takes 106 s instead of 3.6 s (profiles/r05_cpu_probe.txt) - that, not the GPU, is
    # what took the round-4 suite past the driver's 900 s limit.  Small tensors: 4-8 threads is the plateau.
    import torch
    from muscle_amd._host import cpu_share
    torch.set_num_threads(max(1, min(8, cpu_share())))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both_arith: golden / oracle parity test that runs in both GEMM arithmetics by default "
                                       "(exact-fp32 MFMA and the split-bf16 mode of mx_set_gemm_mode(1))")


def pytest_addoption(parser):
    parser.addoption("--arith", default=os.environ.get("MUSCLE_TEST_ARITH", "marked"),
                     choices=["marked", "fp32", "split", "both"],
                     help="GEMM arithmetic of the GPU tests: 'marked' (default) = fp32 everywhere + split for the tests marked "
                          "both_arith; 'fp32' / 'split' = every test in that mode only; 'both' = every test in both modes")


# Every GPU test takes this (autouse) fixture.  Its parameter is the mode handed to muscle_amd.set_gemm_mode() for the
# duration of the test: 0 = exact-fp32 MFMA, 1 = the split arithmetic for the MFMA-bound shapes.
@pytest.fixture(autouse=True)
def gemm_arith(request):
    mode = getattr(request, "param", 0)
    if "gpu" not in request.keywords or not _has_gpu():
        yield 0
        return
    import muscle_amd
    before = muscle_amd.get_gemm_mode()        # the library's default is 1 (split); a test without a parameter runs in exact fp32
    muscle_amd.set_gemm_mode(mode)
    try:
        yield mode
    finally:
        muscle_amd.set_gemm_mode(before)


def pytest_generate_tests(metafunc):
    if "gemm_arith" not in metafunc.fixturenames or metafunc.definition.get_closest_marker("gpu") is None:
        return
    opt = metafunc.config.getoption("--arith")
    marked = metafunc.definition.get_closest_marker("both_arith") is not None
    if opt == "both" or (opt == "marked" and marked):
        metafunc.parametrize("gemm_arith", [0, 1], indirect=True, ids=["fp32", "split"])
    elif opt == "split":
        metafunc.parametrize("gemm_arith", [1], indirect=True, ids=["split"])


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Order of the GPU suite (the driver runs `pytest -x`: the first failure ends the run).  Parity against the oracle and the
# reference's fixtures comes first, hot path before the rows either side of it; tests that compare the build with ITSELF
# (graph replay vs eager, two ranks vs one, run-to-run bits, arithmetic modes) and infrastructure come last, so that a
# property failing there can never hide a parity row again.
_ORDER = [
    "test_oracle_golden", "test_cpu_host",                                  # CPU
    "test_gpu_dwfused", "test_gpu_wgrad", "test_gpu_split",                 # kernel units against fp64 (seconds each): depthwise, GEMMs, the split arithmetic's error bounds
    "test_gpu_backbone", "test_gpu_model", "test_gpu_b7_golden", "test_gpu_phase2", "test_gpu_config2",    # SURVEY 8(a) vs oracle / reference
    "test_gpu_decoder", "test_gpu_fullsize",                                # a19/a20, full-size identities
    "test_gpu_infer", "test_gpu_eval", "test_gpu_irn", "test_input_path",   # 8(f) rows
    "test_gpu_graph", "test_gpu_dist", "test_gpu_determinism",              # self-comparisons / infrastructure
]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = item.module.__name__ if item.module else ""
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER) - 3      # unknown modules: ahead of the self-comparisons
    items.sort(key=rank)            # stable: the order inside a module is kept
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
