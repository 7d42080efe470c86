#!/usr/bin/env python3
"""Model files written by the REFERENCE's own dump functions -- the first network-side artefact in this repository that reference code produced
(VERDICT r5, next 2).  Build container only: it imports /root/reference/misc/*.py where they lie; nothing of them is copied, and the script never travels
to the GPU box (the emitted text, committed beside it, is the fixture).

What runs from the reference:
    misc/taiyaki_flipflop5_guppy.py:28-99   small_hex, process_column, cformatM, cformatV, print_lstm, print_convolution     (LSTM5 models, networks.c:218-254)
    misc/taiyaki_rle5.py:28-99              the same functions                                                               (runnie's model, networks.c:364-400)
    misc/taiyaki_flipflop_guppy.py:28-76    small_hex, process_column, cformatM, cformatV, print_gru                         (GRUmod5 model, networks.c:292-324)
What is provided here so that the three modules IMPORT (their lines 15-24):
    `taiyaki.helpers`, `taiyaki.cmdargs.AutoBool / FileExists` (two argparse actions the module-level parser mentions), `taiyaki.layers.DeltaSample` and
    `taiyaki.layers._cudnn_to_guppy_gru`.  taiyaki is a third-party package absent from this image.  Nothing of it computes anything here EXCEPT
    `_cudnn_to_guppy_gru`, which print_gru calls on every tensor: it is the IDENTITY below -- the weights handed in are already in the order the C code
    reads them (z, r, candidate: layers.c:664-715), i.e. what taiyaki's function would return for them.  That one line is not reference code and says so.
What is restated, because the reference has it inline under `if __name__ == '__main__'` (not importable):
    the header / footer lines and the ORDER of the print_* calls (taiyaki_flipflop5_guppy.py:104-164, taiyaki_rle5.py:104-164, taiyaki_flipflop_guppy.py:79-135), and the GRU
    script's single-feature convolution (taiyaki_flipflop_guppy.py:92-103: one cformatM over filterW.reshape(-1, 1) with nr = 4 winlen - 3, cformatV, and
    three #define lines whose names differ from the LSTM script's).  `run_reference_main` below then ALSO executes each script's own `__main__` block, with
    `helpers.load_model` handing it the same network, and insists that its stdout equals the restated sequence byte for byte -- so the committed files are
    what the reference scripts print.

The network fed in: flappie_amd.model.synthetic_model(kind, hidden = 16 (run-length model: 8), seed = 11) as torch.nn.Conv1d / LSTM / GRU / Linear modules.

usage: python tests/golden/make_mdl_fixture.py        (rewrites tests/golden/ref_writer_{lstm5_h16,grumod5_h16,rle5_h8}.mdl)
"""
import argparse
import contextlib
import importlib.util
import io
import os
import runpy
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

REF_MISC = "/root/reference/misc"
HIDDEN, SEED = 16, 11
_network_for_main = None


def _install_taiyaki_names():
    class _Flag(argparse.Action):                      # taiyaki.cmdargs.AutoBool / FileExists: only ever named by the module-level parser
        def __call__(self, parser, namespace, values, option_string=None):
            setattr(namespace, self.dest, values)

    class DeltaSample:                                 # (the scripts only ask isinstance(first layer, DeltaSample))
        pass

    pkg = types.ModuleType("taiyaki")
    helpers = types.ModuleType("taiyaki.helpers")
    helpers.load_model = lambda path: _network_for_main
    cmdargs = types.ModuleType("taiyaki.cmdargs")
    cmdargs.AutoBool, cmdargs.FileExists = _Flag, _Flag
    layers = types.ModuleType("taiyaki.layers")
    layers.DeltaSample = DeltaSample
    layers._cudnn_to_guppy_gru = lambda t: t           # IDENTITY, see the header: NOT reference code
    pkg.helpers, pkg.cmdargs, pkg.layers = helpers, cmdargs, layers
    sys.modules.update({"taiyaki": pkg, "taiyaki.helpers": helpers, "taiyaki.cmdargs": cmdargs, "taiyaki.layers": layers})


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF_MISC, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Box:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _param(a):
    return torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), requires_grad=False)


def torch_network(mdl):
    """the attribute paths the dump scripts walk (network.sublayers[i].conv / .stride / .layer.lstm / .lstm / .cudnn_gru / .linear) over torch modules holding
    the synthetic model's numbers"""
    subs = []
    for c in mdl.convs:
        conv = torch.nn.Conv1d(c.nf, c.W.nc, c.winlen, stride=c.stride)
        conv.weight = _param(c.taps().transpose(0, 2, 1))          # [nfilter, winlen, nf] -> torch's [nfilter, nf, winlen]
        conv.bias = _param(c.b.dense().reshape(-1))
        subs.append(_Box(conv=conv, stride=c.stride))
    H = mdl.hidden
    for i, r in enumerate(mdl.rnns):
        if mdl.kind == M.NET_GRUMOD5:
            cell = torch.nn.GRU(r.iW.nr, H)
            holder = _Box(cudnn_gru=cell)
        else:
            cell = torch.nn.LSTM(r.iW.nr, H)
            holder = _Box(lstm=cell)
        cell.weight_ih_l0 = _param(r.iW.dense())                   # flappie's [nc = G H][nr = in] image IS torch's [G H, in]
        cell.weight_hh_l0 = _param(r.sW.dense())
        cell.bias_ih_l0 = _param(r.b.dense().reshape(-1))
        subs.append(_Box(layer=holder) if i % 2 == 0 else holder)  # layers 1, 3, 5 run backward: wrapped in taiyaki's Reverse (`.layer`)
    lin = torch.nn.Linear(H, mdl.nparam)
    lin.weight = _param(mdl.FF_W.dense())
    lin.bias = _param(mdl.FF_b.dense().reshape(-1))
    subs.append(_Box(linear=lin))
    return _Box(sublayers=subs)


def lstm5_text(ref, net, ident, fam="flipflop5", guard="FLIPFLOP"):
    """taiyaki_flipflop5_guppy.py:104-164 (taiyaki_rle5.py:104-164 with fam = "rle5", guard = "RLE"), the calls in the script's order"""
    modelid = ident + "_"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sys.stdout.write("""#pragma once
    #ifndef {g}_{}MODEL_H
    #define {g}_{}MODEL_H
    #include "../util.h"
    """.format(modelid.upper(), modelid.upper(), g=guard))
        for i in (1, 2, 3):
            ref.print_convolution(net.sublayers[i - 1], "conv{}_rnnrf_{}_{}".format(i, fam, modelid), scale=False)
        ref.print_lstm(net.sublayers[3].layer, "lstmB1_rnnrf_{}_{}".format(fam, modelid))
        ref.print_lstm(net.sublayers[4], "lstmF2_rnnrf_{}_{}".format(fam, modelid))
        ref.print_lstm(net.sublayers[5].layer, "lstmB3_rnnrf_{}_{}".format(fam, modelid))
        ref.print_lstm(net.sublayers[6], "lstmF4_rnnrf_{}_{}".format(fam, modelid))
        ref.print_lstm(net.sublayers[7].layer, "lstmB5_rnnrf_{}_{}".format(fam, modelid))
        gn = net.sublayers[8]
        ref.cformatM(sys.stdout, "FF_rnnrf_{}_{}W".format(fam, modelid), gn.linear.weight)
        ref.cformatV(sys.stdout, "FF_rnnrf_{}_{}b".format(fam, modelid), gn.linear.bias)
        sys.stdout.write("#endif /* {}_{}MODEL_H */".format(guard, modelid.upper()))
    return buf.getvalue()


def rle5_text(ref, net, ident):
    return lstm5_text(ref, net, ident, "rle5", "RLE")


def grumod5_text(ref, net, ident):
    """taiyaki_flipflop_guppy.py:79-135"""
    modelid = ident + "_"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sys.stdout.write("""#pragma once
    #ifndef FLIPFLOP_{}MODEL_H
    #define FLIPFLOP_{}MODEL_H
    #include "../util.h"
    """.format(modelid.upper(), modelid.upper()))
        filterW = net.sublayers[0].conv.weight
        nfilter, _, winlen = filterW.shape
        ref.cformatM(sys.stdout, "conv_rnnrf_flipflop_{}W".format(modelid), filterW.reshape(-1, 1), nr=winlen * 4 - 3, nc=nfilter)
        ref.cformatV(sys.stdout, "conv_rnnrf_flipflop_{}b".format(modelid), net.sublayers[0].conv.bias.reshape(-1))
        sys.stdout.write("#define conv_rnnrf_flipflop_{}stride  {}\n".format(modelid, net.sublayers[0].stride))
        sys.stdout.write("""#define {}nfilter  {}
    #define _conv_rnnrf_flipflop_{}winlen  {}
    """.format(modelid, nfilter, modelid, winlen))
        ref.print_gru(net.sublayers[1].layer, "gruB1_rnnrf_flipflop_{}".format(modelid))
        ref.print_gru(net.sublayers[2], "gruF2_rnnrf_flipflop_{}".format(modelid))
        ref.print_gru(net.sublayers[3].layer, "gruB3_rnnrf_flipflop_{}".format(modelid))
        ref.print_gru(net.sublayers[4], "gruF4_rnnrf_flipflop_{}".format(modelid))
        ref.print_gru(net.sublayers[5].layer, "gruB5_rnnrf_flipflop_{}".format(modelid))
        gn = net.sublayers[6]
        ref.cformatM(sys.stdout, "FF_rnnrf_flipflop_{}W".format(modelid), gn.linear.weight)
        ref.cformatV(sys.stdout, "FF_rnnrf_flipflop_{}b".format(modelid), gn.linear.bias)
        sys.stdout.write("#endif /* FLIPFLOP_{}MODEL_H */".format(modelid.upper()))
    return buf.getvalue()


def run_reference_main(script, net, ident):
    """the script's own `__main__` block, fed `net` through helpers.load_model: what `taiyaki_flipflop*_guppy.py --id IDENT model.checkpoint` prints"""
    global _network_for_main
    _network_for_main = net
    argv, sys.argv = sys.argv, [script, "--id", ident, "model.checkpoint"]
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(REF_MISC, script + ".py"), run_name="__main__")
    finally:
        sys.argv = argv
    return buf.getvalue()


def main():
    if not os.path.isdir(REF_MISC):
        sys.exit("the reference checkout is not here: the committed .mdl files are the fixture")
    _install_taiyaki_names()
    jobs = ((M.NET_LSTM5, "r941native", "taiyaki_flipflop5_guppy", lstm5_text, "ref_writer_lstm5_h16.mdl", HIDDEN),
            (M.NET_GRUMOD5, "r941native5mC", "taiyaki_flipflop_guppy", grumod5_text, "ref_writer_grumod5_h16.mdl", HIDDEN),
            (M.NET_LSTM5_RLE, "r941native", "taiyaki_rle5", rle5_text, "ref_writer_rle5_h8.mdl", 8))
    for kind, ident, script, restated, out, hidden in jobs:
        mdl = M.synthetic_model(kind, hidden, seed=SEED, ident=ident)
        net = torch_network(mdl)
        ref = _load(script)
        text = restated(ref, net, ident)
        own = run_reference_main(script, net, ident)
        assert own == text, "%s: the script's own __main__ prints something else than the restated call sequence" % script
        with open(os.path.join(HERE, out), "w") as fh:
            fh.write(text)
        print("%s: %d bytes from %s (functions imported; the script's own __main__ prints the same bytes)" % (out, len(text), script))


if __name__ == "__main__":
    main()
