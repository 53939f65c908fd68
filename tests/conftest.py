import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The driver gives `pytest -m gpu` 1200 s on the GPU box (GPUTEST_r03.json: steps[0].timeout_s); a run killed at that limit
# counts as a failed suite.  Tests whose fp32 oracle loop on the device takes tens of seconds to minutes carry
# `@pytest.mark.cost(seconds)` and run FIRST, the most expensive first: what validates the defaults (the headline 40-step case
# first of all) runs before anything can eat its time, and a box too slow for the suite shows up as the suite's timeout, not as
# thinner coverage.  Only the cases marked `cost(seconds, optional=True)` (second seeds / second models) may be skipped for the
# budget, and every such skip is repeated LOUDLY in the terminal summary; DWM_STRICT_BUDGET=1 turns it into a failure.
# DWM_HEAVY_TESTS=1 lifts the budget and adds the extended cases (more seeds / frames / views, tests.common.HEAVY); their recorded
# results are under profiles/.
SUITE_T0 = time.time()
SUITE_BUDGET_S = float(os.environ.get("DWM_SUITE_BUDGET_S", "1050"))
# what the ~430 tests without a cost mark take after the cost-marked ones: 360 s in the full run of round 5 (647 s in all with the
# tVAE window test skipped, profiles/r5z_pytest.log); 400 leaves that test room on such a box (~850 s in all) and drops it on a slower one
REST_OF_SUITE_S = float(os.environ.get("DWM_SUITE_REST_S", "400"))
BUDGET_SKIPS = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")
    config.addinivalue_line("markers", "cost(seconds, optional=False, priority=5): measured duration of a long GPU test (see SUITE_BUDGET_S)")
    config.addinivalue_line("markers", "quick: the kernel batteries + small-model forwards (`pytest -m 'gpu and quick'`: ~1 GPU-minute) - "
                                       "what a builder runs between kernel changes; the driver runs everything")


def pytest_runtest_setup(item):
    m = item.get_closest_marker("cost")
    if m is None or not m.kwargs.get("optional") or os.environ.get("DWM_HEAVY_TESTS"):
        return
    used = time.time() - SUITE_T0
    if used + float(m.args[0]) + REST_OF_SUITE_S > SUITE_BUDGET_S:
        msg = (f"suite time budget: {used:.0f} s used + ~{m.args[0]} s for this test + ~{REST_OF_SUITE_S:.0f} s for the rest of the suite "
               f"> {SUITE_BUDGET_S:.0f} s (DWM_HEAVY_TESTS=1 runs it regardless)")
        if os.environ.get("DWM_STRICT_BUDGET"):
            pytest.fail("optional cost-marked test would be skipped and DWM_STRICT_BUDGET is set: " + msg)
        BUDGET_SKIPS.append(item.nodeid)
        pytest.skip(msg)


def pytest_terminal_summary(terminalreporter):
    if BUDGET_SKIPS:
        terminalreporter.section("COST-MARKED TESTS SKIPPED FOR THE SUITE TIME BUDGET", sep="!")
        for n in BUDGET_SKIPS:
            terminalreporter.write_line("  SKIPPED (budget): " + n)
        terminalreporter.write_line("  a green run with this section did NOT execute these checks (DWM_STRICT_BUDGET=1 makes it a failure)")


@pytest.fixture(scope="session")
def small_cfg():
    from tests.common import small_config
    return small_config()


def pytest_collection_modifyitems(config, items):
    """the GPU leg calls through the C ABI of the in-tree libdwm_hip.so: build it once if the snapshot came without it"""
    if any("gpu" in item.keywords for item in items) and config.getoption("-m") != "not gpu":
        try:
            from opendwm_amd.build import ensure_built
            ensure_built()
        except Exception as e:                          # the tests themselves then fail loudly in _lib.load()
            print(f"[conftest] libdwm_hip.so could not be built: {e}", file=sys.stderr)
    # `quick`: every GPU test of the kernel-level files that carries no cost mark
    quick_files = ("test_hip_gpu.py", "test_gemm4w_gpu.py", "test_round5_kernels_gpu.py", "test_stream32_gpu.py")
    for it in items:
        if "gpu" in it.keywords and it.get_closest_marker("cost") is None and os.path.basename(str(it.fspath)) in quick_files:
            it.add_marker(pytest.mark.quick)
    # long tests first, required ones before optional ones, the most expensive first (stable: the others keep their order)
    # (optional cases in the order of `priority` - the class-default cached-adapter mode first -, then the most expensive first)
    def rank(it):
        m = it.get_closest_marker("cost")
        if m is None:
            return (2, 0, 0.0)
        return (1, int(m.kwargs.get("priority", 5)), -float(m.args[0])) if m.kwargs.get("optional") else (0, 0, -float(m.args[0]))
    items.sort(key=rank)
