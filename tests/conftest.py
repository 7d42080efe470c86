import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def engine():
    """HIP engine on device 0.  Fails loudly (no CPU fallback) when the library or the GPU is missing."""
    from flappie_amd import binding as B
    eng = B.Engine(0)
    yield eng
    eng.close()


def oracle_calls(mdl, sigs, workers=None, **kw):
    """The oracle's basecall of every signal, on a thread pool (ctypes releases the interpreter lock, the oracle keeps no state in
    dot mode 0; threads, not processes: nothing of a live HIP context is forked).  TEST INFRASTRUCTURE."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ffo
    om = ffo.OracleModel(mdl)
    workers = workers or min(64, os.cpu_count() or 1)
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(lambda s: om.basecall(s, **kw) if len(s) else None, sigs))


# ---- what the GPU parity tests saw, printed at the end of a run (the bounds they assert are in the test files; this is the margin)
PARITY_SEEN = {"dtrans": 0.0, "dpost": 0.0, "reads": 0}


def note_parity(dtrans, dpost=None):
    PARITY_SEEN["dtrans"] = max(PARITY_SEEN["dtrans"], float(dtrans))
    if dpost is not None:
        PARITY_SEEN["dpost"] = max(PARITY_SEEN["dpost"], float(dpost))
    PARITY_SEEN["reads"] += 1


def pytest_terminal_summary(terminalreporter):
    if PARITY_SEEN["reads"]:
        terminalreporter.write_line("parity seen in this run: %d reads compared with the oracle, worst |dtrans| %.3e (bound 5e-5; north_star 1e-4), worst end-to-end |dlogpost| %.3e (bound 1e-4)"
                                    % (PARITY_SEEN["reads"], PARITY_SEEN["dtrans"], PARITY_SEEN["dpost"]))
