"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/ffhip.h
declares (no compute is called without a GPU), the .mdl text format round-trips, and the product
package never touches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from flappie_amd import model as M

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;{]*;", "", text)            # function-pointer typedefs are not symbols
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if not n.startswith("__")))


def test_library_exports_every_declared_symbol():
    from flappie_amd import binding as B
    path = B.library_path()
    assert os.path.exists(path), "libffhip.so not built: run python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(path)
    declared = _declared_functions("ffhip.h")
    assert len(declared) >= 25
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, "symbols declared in include/ffhip.h but not exported: %s" % missing
    assert b"gfx950" in C.cast(L.ffhip_version, C.CFUNCTYPE(C.c_char_p))()


def test_host_library_exports_reference_api():
    """The C host layer keeps the reference's names (networks.h:31-42, decode.h:21-38, flappie_matrix.h:39-61)."""
    path = os.path.join(ROOT, "flappie_amd", "libflappie_host.so")
    if not os.path.exists(path):
        pytest.skip("host layer not built yet")
    L = C.CDLL(path)
    for hdr in ("flappie_matrix.h", "flappie_structures.h", "networks.h", "decode.h", "flappie_common.h", "layers.h", "flappie_output.h"):
        if not os.path.exists(os.path.join(ROOT, "include", hdr)):
            continue
        missing = [n for n in _declared_functions(hdr) if not hasattr(L, n)]
        assert not missing, "%s: not exported: %s" % (hdr, missing)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device the engine must refuse to exist -- there is no CPU path in the product."""
    from flappie_amd import binding as B
    if B.lib().ffhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(B.FFHipError):
        B.Engine(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flappie_amd")):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "ff_oracle" not in text and "import ffo" not in text and "from oracle" not in text, fn


def test_release_library_has_no_test_hook_and_few_switches():
    """VERDICT r4 weak 12 / ADVICE r4: the host-load rehearsal hook (placeholder basecalls on an environment variable) lives in the
    -DFFHIP_TEST_HOOKS build only (tools/test_hooks/libffhip.so, `make hooks`), and the release build reads at most 15 environment variables."""
    import re
    blob = open(os.path.join(ROOT, "flappie_amd", "libffhip.so"), "rb").read()
    assert b"REHEARSAL" not in blob
    names = set()
    for lib in ("libffhip.so", "libflappie_host.so", "flappie"):
        path = os.path.join(ROOT, "flappie_amd", lib)
        if os.path.exists(path):
            names |= set(re.findall(rb"(?:FFHIP|FLAPPIE)_[A-Z][A-Z0-9_]+", open(path, "rb").read()))
    src_vars = set()
    for sub in ("csrc", "host"):
        for fn in os.listdir(os.path.join(ROOT, "flappie_amd", sub)):
            if fn.endswith((".hip", ".hpp", ".c", ".h")):
                text = open(os.path.join(ROOT, "flappie_amd", sub, fn)).read()
                text = re.sub(r"#ifdef FFHIP_TEST_HOOKS.*?#e(?:lse|ndif)", "", text, flags=re.S)
                src_vars |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', text))
    assert len(src_vars) <= 15, sorted(src_vars)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for v in src_vars:
        assert v in doc, "undocumented environment variable " + v
    hooks = os.path.join(ROOT, "tools", "test_hooks", "libffhip.so")
    if os.path.exists(hooks):
        assert b"FFHIP_DEBUG_HOST_REHEARSAL_MSPS" in open(hooks, "rb").read()


@pytest.mark.parametrize("kind,hidden", [(M.NET_LSTM5, 32), (M.NET_GRUMOD5, 48)])
def test_mdl_roundtrip(tmp_path, kind, hidden):
    mdl = M.synthetic_model(kind, hidden, seed=3, ident="r941native" if kind == M.NET_LSTM5 else "r941native5mC")
    path = str(tmp_path / "model.mdl")
    M.write_mdl(path, mdl)
    text = open(path).read()
    # the grammar networks.c:218-323 relies on
    prefix = "conv1_rnnrf_flipflop5_r941native_" if kind == M.NET_LSTM5 else "conv_rnnrf_flipflop_r941native5mC_"
    assert "_Mat _%sW = {" % prefix in text
    assert "#define %sstride  " % prefix in text
    assert "const flappie_matrix %sW = &_%sW;" % (prefix, prefix) in text
    back = M.load_mdl(path, kind, mdl.ident)
    assert back.hidden == hidden and back.nparam == mdl.nparam
    for a, b in zip(mdl.convs, back.convs):
        assert (a.stride, a.winlen, a.nf) == (b.stride, b.winlen, b.nf)
        assert np.array_equal(a.W.data, b.W.data) and np.array_equal(a.b.data, b.b.data)
    for a, b in zip(mdl.rnns, back.rnns):
        assert np.array_equal(a.iW.data, b.iW.data) and np.array_equal(a.sW.data, b.sW.data)
        assert np.array_equal(a.b.data, b.b.data)
    assert np.array_equal(mdl.FF_W.data, back.FF_W.data)


def test_flop_count_matches_baseline_formula():
    # 80 H^2 + 688 H + 3400 per block for the LSTM5 architecture
    for H in (256, 384, 512):
        mdl = M.synthetic_model(M.NET_LSTM5, H, seed=1)
        assert mdl.flop_per_block() == 80 * H * H + 688 * H + 3400


def test_split_operand_packer_fp16_rounding(tmp_path):
    """The host half of the split operand format (flappie_amd/csrc/ffhip_split.hpp packs the weights as fp16 slices): its
    fp32 -> fp16 conversion must round to nearest even exactly as IEEE does (numpy float16), subnormals and overflow included."""
    import subprocess
    src = tmp_path / "h.cpp"
    src.write_text('#include "ffhip_split.hpp"\n#include <cstdio>\nint main() { float f; while (fread(&f, 4, 1, stdin) == 1) {'
                   ' uint16_t h = ffhip::split_host_f16_rne(f); float b = ffhip::split_host_f16_value(h); fwrite(&h, 2, 1, stdout); fwrite(&b, 4, 1, stdout); } return 0; }\n')
    exe = str(tmp_path / "h")
    subprocess.check_call(["g++", "-O2", "-I", os.path.join(ROOT, "flappie_amd", "csrc"), "-o", exe, str(src)])
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(100000) * rng.choice([1e-8, 1e-6, 1e-4, 1, 100, 30000, 70000], 100000),
                        [0, -0.0, 65504, 65519.9, 65520, 65536, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.99e-8, 6.1e-5, 6.0975e-5, np.inf, -np.inf]]).astype(np.float32)
    out = subprocess.run([exe], input=x.tobytes(), capture_output=True, check=True).stdout
    rec = np.frombuffer(out, dtype=np.dtype([("h", "<u2"), ("b", "<f4")]))
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16)
    assert np.array_equal(rec["h"], ref.view(np.uint16))
    assert np.array_equal(rec["b"], ref.astype(np.float32))


def kmer_count(s, k=5):
    return len({s[i:i + k] for i in range(len(s) - k + 1)})


@pytest.mark.parametrize("kind,hidden,seed", [(M.NET_LSTM5, 64, 7), (M.NET_LSTM5, 128, 7), (M.NET_LSTM5, 256, 1), (M.NET_LSTM5, 384, 1), (M.NET_LSTM5, 384, 5),
                                              (M.NET_LSTM5, 512, 1), (M.NET_GRUMOD5, 64, 7), (M.NET_GRUMOD5, 256, 1), (M.NET_GRUMOD5, 256, 3), (M.NET_LSTM5, 256, 3)])
def test_synthetic_models_are_input_driven(kind, hidden, seed):
    """The random models the parity tests and bench.py use must make the comparison mean something: on N(0,1)
    reads of 4000 samples the oracle calls at least one base per 12 samples and at least 100 distinct 5-mers per read (input-driven,
    no period), and the network is not chaotic -- a 1e-6 perturbation of the input moves the transition scores by less than 5e-5
    (flappie_amd/model.py SYNTH_GAINS; searched with tools/dev/tune_gains.py).  The oracle's vectorised mode: this is about the model."""
    from oracle import ffo
    mdl = M.synthetic_model(kind, hidden, seed=seed)
    om = ffo.OracleModel(mdl)
    rng = np.random.default_rng(1234)
    with ffo.dot_mode(2):
        for _ in range(2):
            sig = rng.standard_normal(4000).astype(np.float32)
            a = om.basecall(sig)
            other = om.transitions(sig + np.float32(1e-6) * rng.standard_normal(4000).astype(np.float32))
            calls = a["basecall"]
            assert len(calls) * 12 >= 4000, (len(calls), calls[:60])
            assert kmer_count(calls) >= 100, (kmer_count(calls), calls[:60])
            assert min(calls.count(c) for c in "ACGTZ"[: mdl.nbase]) >= 0.02 * len(calls)
            stays = int(np.sum(a["path"][1:] == a["path"][:-1]))
            assert 0 < stays < a["nblock"]                       # stays and moves both occur
            assert float(np.abs(other - a["trans"]).max()) < 5e-5
