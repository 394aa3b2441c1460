import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fp64_bit_identity: pins bits between planner bodies / encoder forms of the default FP64 path; skipped "
                                       "when the suite runs with --nlc-planner-opt gru_gemm=1")
    config.addinivalue_line("markers", "no_sliced_instance: an encoder_mode test whose models have no int8-sliced encoder instance "
                                       "(hidden_units other than 128): the option must be harmless, no sliced launch is expected")


def pytest_collection_modifyitems(config, items):
    """`--nlc-planner-opt gru_gemm=1` runs the whole GPU suite in the sliced mode.  The tests marked `fp64_bit_identity` pin
    BITS between bodies that then run different encoders -- the two-launch bodies the int8-sliced kernel, the fused one-launch
    body and the cooperative form the FP64 one (include/nlc.h, "gru_gemm") -- or assert that a default planner does not take the
    option: they are skipped in that mode, with this reason, and run in the default one."""
    opts = _parse_opts(config.getoption("--nlc-planner-opt"))
    if opts.get("gru_gemm") == 1.0:
        skip = pytest.mark.skip(reason="pins bit-identity between bodies that run different encoders under gru_gemm = 1")
        for item in items:
            if item.get_closest_marker("fp64_bit_identity"):
                item.add_marker(skip)


def pytest_addoption(parser):
    parser.addoption(
        "--nlc-planner-opt", action="append", default=[], metavar="NAME=VALUE",
        help="run the WHOLE GPU suite with this nlc_set_option default on every planner / model ctx the tests build "
             "(neurallaplacecontrol_amd.set_default_options), e.g. --nlc-planner-opt gru_gemm=1 for the int8-sliced encoder; "
             "tests that pin bit-identity between the FP64 bodies relax to ENCODER_MODE_TOL in that mode (gpu_common)")


def _parse_opts(pairs):
    out = {}
    for kv in pairs:
        name, _, value = kv.partition("=")
        out[name] = float(value)
    return out


@pytest.fixture(scope="session", autouse=True)
def _suite_wide_planner_options(request):
    opts = _parse_opts(request.config.getoption("--nlc-planner-opt"))
    if not opts:
        yield {}
        return
    from neurallaplacecontrol_amd import set_default_options

    old = set_default_options(opts)
    yield opts
    set_default_options(old)


# what the fast mode needs besides the option for the sliced kernel to be the one that runs at fixture sizes: the encoder launch in
# its wave-sized form (auto picks the cooperative FP64 form up to 50 000 windows) and the two-launch planner bodies (the fused
# one-launch body keeps its FP64 encoder role).  A test's own planner_options still win.
SLICED_MODE_OPTIONS = {"gru_gemm": 1.0, "gru_coop": 0.0, "fused_max_samples": 0.0}


@pytest.fixture(params=["fp64", "i8sliced"])
def encoder_mode(request, _suite_wide_planner_options):
    """The NL golden / full-size tests run once per encoder: the default FP64-MFMA GRU and the labelled fast mode `gru_gemm = 1`
    (hidden-state GEMMs as int8-sliced 54-bit fixed point, csrc/kernels_gru_i8.hip) -- the same fixtures at the same tolerances,
    so the driver's GPU record shows both.  Every ctx created inside the test gets the options; at teardown
    the library's launch counter must say that the sliced kernel ran in the fast mode (and did not in the default one)."""
    from neurallaplacecontrol_amd import _lib, set_default_options

    opts = dict(_suite_wide_planner_options)
    if request.param == "i8sliced":
        opts.update(SLICED_MODE_OPTIONS)
    else:
        opts["gru_gemm"] = 0.0
    old = set_default_options(opts)
    probe = _lib.Ctx(0)
    before = probe.get_stat("gru_i8_launches")
    yield request.param
    ran = probe.get_stat("gru_i8_launches") - before
    set_default_options(old)
    if request.param == "fp64" or request.node.get_closest_marker("no_sliced_instance"):
        assert ran == 0, f"{ran} int8-sliced encoder launches in a test that must not take the option"
    else:
        assert ran > 0, "encoder_mode = i8sliced, but no launch took the int8-sliced encoder kernel"


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def nlc():
    """The product package, on a GPU box (the -m gpu tests): a missing device is a failure, not a skip."""
    import torch

    import neurallaplacecontrol_amd as n

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return n
