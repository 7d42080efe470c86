"""Model containers for the flip-flop basecaller: synthetic weights and the `.mdl` text format.

The shipped flappie models are C headers (`src/models/*.mdl`) that define, per tensor NAME,
``float __NAME[]``, ``_Mat _NAME = {.nr,.nrq,.nc,.stride,.data.f}`` and ``const flappie_matrix NAME``
plus ``#define <conv prefix>stride N`` (format defined by the reference's dump scripts,
misc/taiyaki_flipflop5_guppy.py:38-99 and misc/taiyaki_flipflop_guppy.py:92-133; consumed by
networks.c:218-399).  In the reference checkout every .mdl is a git-LFS pointer stub, so this module
can (a) write that exact text format from arrays, (b) parse it back, and (c) generate seeded synthetic
models with the architecture of the registry entries.

A matrix is stored as a float32 array of shape [nc, stride] (flappie's column-major [nr x nc] with
each column zero-padded to a multiple of 4 floats, flappie_matrix.h:18-24).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

NET_LSTM5 = 0    # flipflop5_guppy_transitions, networks.c:539-586 (r941_native, r941_rna002, r103_native)
NET_GRUMOD5 = 1  # flipflop_guppy_transitions,  networks.c:450-489 (r941_5mC)
NET_LSTM5_RLE = 2  # runlength5_guppy_transitions, networks.c:672-725 (runnie's rle_r941_native): LSTM5 trunk, run-length head

# registry: networks.h:18-26, networks.c:21-105.  `ident` is the <id> in the tensor names.
REGISTRY = {
    "r941_native": dict(kind=NET_LSTM5, ident="r941native", enum=0,
                        description="R9.4.1 model for MinION.  Trained from native DNA library"),
    "r941_rna002": dict(kind=NET_LSTM5, ident="r941rna002", enum=1,
                        description="R9.4.1 dRNA model for MinION.  Trained from native and synthetic RNA library"),
    "r941_5mC": dict(kind=NET_GRUMOD5, ident="r941native5mC", enum=2,
                     description="R9.4.1 model for PromethION; 5mC aware.  Trained from native NA12878 library"),
    "r103_native": dict(kind=NET_LSTM5, ident="r103native", enum=3,
                        description="R10.3 model for MinION.  Trained from native DNA library"),
    "rle_r941_native": dict(kind=NET_LSTM5_RLE, ident="r941native", enum=5,
                            description="R9.4.1 run-length encoded model for MinION.  Trained from native DNA library"),
}
# hidden sizes inferred from the LFS stub byte counts; never hard-coded in kernels
INFERRED_HIDDEN = {"r941_native": 384, "r941_rna002": 384, "r941_5mC": 256, "r103_native": 512}


@dataclass
class Mat:
    """flappie matrix: logical [nr x nc], storage data[nc, stride]."""
    nr: int
    nc: int
    data: np.ndarray

    @property
    def nrq(self) -> int:
        return (self.nr + 3) // 4

    @property
    def stride(self) -> int:
        return 4 * self.nrq

    @staticmethod
    def from_dense(a: np.ndarray) -> "Mat":
        """a[nc, nr] (PyTorch [out][in]) -> padded."""
        a = np.asarray(a, dtype=np.float32)
        nc, nr = a.shape
        stride = 4 * ((nr + 3) // 4)
        d = np.zeros((nc, stride), dtype=np.float32)
        d[:, :nr] = a
        return Mat(nr, nc, d)

    @staticmethod
    def vector(v: np.ndarray) -> "Mat":
        v = np.asarray(v, dtype=np.float32).reshape(1, -1)
        return Mat.from_dense(v)

    def dense(self) -> np.ndarray:
        return self.data[:, : self.nr]


@dataclass
class ConvLayer:
    W: Mat        # [nf_pad*winlen - nf_pad + nf  x  nfilter]
    b: Mat        # [nfilter x 1]
    stride: int
    nf: int       # input features
    winlen: int

    def taps(self) -> np.ndarray:
        """[nfilter, winlen, nf] dense filter taps."""
        nf_pad = 4 * ((self.nf + 3) // 4)
        out = np.zeros((self.W.nc, self.winlen, self.nf), dtype=np.float32)
        for w in range(self.winlen):
            out[:, w, :] = self.W.data[:, w * nf_pad: w * nf_pad + self.nf]
        return out


@dataclass
class RnnLayer:
    iW: Mat       # [H_in x G*H]
    sW: Mat       # [H x G*H]
    b: Mat        # [G*H]


@dataclass
class FlipflopModel:
    kind: int
    convs: List[ConvLayer]
    rnns: List[RnnLayer]
    FF_W: Mat
    FF_b: Mat
    ident: str = "synthetic"

    @property
    def hidden(self) -> int:
        return self.rnns[0].sW.nr

    @property
    def ngate(self) -> int:
        return 3 if self.kind == NET_GRUMOD5 else 4

    @property
    def nparam(self) -> int:
        return self.FF_W.nc

    @property
    def nbase(self) -> int:
        return int(round((-1.0 + math.sqrt(1 + 2 * self.nparam)) / 2.0))

    @property
    def nstate(self) -> int:
        return 2 * self.nbase

    @property
    def total_stride(self) -> int:
        s = 1
        for c in self.convs:
            s *= c.stride
        return s

    def nblock(self, nsample: int) -> int:
        n = nsample
        for c in self.convs:
            n = (n + c.stride - 1) // c.stride
        return n

    def flop_per_block(self) -> int:
        """Algorithmic FLOPs per output block."""
        H = self.hidden
        fl = 0
        blocks_per = self.total_stride
        for c in self.convs:
            blocks_per //= c.stride
            fl += 2 * c.nf * c.winlen * c.W.nc * blocks_per
        G = self.ngate
        fl += 5 * (2 * G * H * H + 2 * G * H * H)
        fl += 2 * H * self.nparam
        return fl


def _conv_mat(rng, nf, nfilter, winlen) -> Mat:
    nf_pad = 4 * ((nf + 3) // 4)
    nr = nf_pad * winlen - nf_pad + nf
    a = 1.0 / math.sqrt(nf * winlen)
    taps = rng.uniform(-a, a, size=(nfilter, winlen, nf)).astype(np.float32)
    dense = np.zeros((nfilter, nr), dtype=np.float32)
    for w in range(winlen):
        dense[:, w * nf_pad: w * nf_pad + nf] = taps[:, w, :]
    return Mat.from_dense(dense)


# Gains applied on top of the U(-1/sqrt(fan_in), 1/sqrt(fan_in)) draw: (input weights, recurrent weights, head weights,
# bias added to the "stay" transitions, convolution weights).  With unit gains a random stack is input-blind: three
# swish convolutions shrink an N(0,1) signal to a standard deviation of 0.02 per feature, the recurrent layers then
# see their own fixed point and every read decodes to one periodic string.  The values below were searched
# (tools/dev/tune_gains.py, oracle on N(0,1) reads of 4000 samples, seeds 1 2 3 5 7 102, H = 64 ... 512) for
#   * >= 1 called base per 12 samples and >= 100 distinct 5-mers per 4000-sample read (input-driven: stays and moves,
#     all bases, no period), and
#   * non-chaotic: a 1e-6 perturbation of the input moves the transition scores by < 5e-5,
# and tests/test_cabi_and_model.py::test_synthetic_models_are_input_driven holds the models the parity tests use to it.
# The convolution gain keeps the signal's amplitude through the three thin layers (0.2 ... 1 per feature at the first
# recurrent layer); a moderate head gain keeps the scores off tanh's rails, so qualities vary as well.
SYNTH_GAINS = {NET_LSTM5: (5.0, 2.0, 2.0, -0.3, 3.0), NET_GRUMOD5: (2.0, 1.0, 4.0, 0.0, 2.0), NET_LSTM5_RLE: (5.0, 2.0, 2.0, 0.0, 3.0)}
# rounds 1-3 (input-blind at every size on white-noise reads; kept for the recorded reads of tests/golden/fuzz_tail.npz,
# which were found on synthetic_model(NET_LSTM5, 256, seed=102) under these)
SYNTH_GAINS_R3 = {NET_LSTM5: (5.0, 2.5, 10.0, 0.6), NET_GRUMOD5: (2.0, 1.0, 4.0, 0.3), NET_LSTM5_RLE: (5.0, 2.5, 10.0, 0.6)}


def synthetic_model(kind: int = NET_LSTM5, hidden: int = 384, seed: int = 1,
                    ident: str = "synthetic", gains=None) -> FlipflopModel:
    """Seeded model with the registry architectures' dimensions:
    weights U(-a,a) with a = gain/sqrt(fan_in), biases U(-0.1,0.1) (+ stay bias on the head).
    `gains` = (input, recurrent, head, stay bias[, convolution]); default SYNTH_GAINS[kind]."""
    rng = np.random.default_rng(seed)
    H = hidden

    def bias(n):
        return Mat.vector(rng.uniform(-0.1, 0.1, size=n).astype(np.float32))

    def dense(nin, nout):
        a = 1.0 / math.sqrt(nin)
        return Mat.from_dense(rng.uniform(-a, a, size=(nout, nin)).astype(np.float32))

    if kind in (NET_LSTM5, NET_LSTM5_RLE):
        convs = [ConvLayer(_conv_mat(rng, 1, 4, 5), bias(4), 1, 1, 5),
                 ConvLayer(_conv_mat(rng, 4, 16, 5), bias(16), 1, 4, 5),
                 ConvLayer(_conv_mat(rng, 16, H, 19), bias(H), 5, 16, 19)]
        G, nbase = 4, 4
    elif kind == NET_GRUMOD5:
        convs = [ConvLayer(_conv_mat(rng, 1, H, 19), bias(H), 2, 1, 19)]
        G, nbase = 3, 5
    else:
        raise ValueError("unknown network kind")
    g = tuple(gains) if gains is not None else SYNTH_GAINS[kind]
    gi, gs, gf, stay = g[:4]
    gc = g[4] if len(g) > 4 else 1.0
    if gc != 1.0:
        for c in convs:
            c.W.data *= np.float32(gc)
    rnns = []
    for _ in range(5):
        iW, sW, b = dense(H, G * H), dense(H, G * H), bias(G * H)
        iW.data *= np.float32(gi)
        sW.data *= np.float32(gs)
        rnns.append(RnnLayer(iW, sW, b))
    P = 2 * nbase * (nbase + 1)
    nstate = 2 * nbase
    FF_W, FF_b = dense(H, P), bias(P)
    FF_W.data *= np.float32(gf)
    if kind == NET_LSTM5_RLE:                     # stay -> stay transitions of the run-length head (layers.c:1241-1246)
        for b in range(nbase):
            FF_b.data[0, nstate + b * nstate + b + nbase] += np.float32(stay)
    else:
        for to in range(nbase):                   # flip self-transitions
            FF_b.data[0, to * nstate + to] += np.float32(stay)
        for b2 in range(nbase, nstate):           # flop stays
            FF_b.data[0, nbase * nstate + b2] += np.float32(stay)
    return FlipflopModel(kind, convs, rnns, FF_W, FF_b, ident)


# --------------------------------------------------------------------------- .mdl text format

_TRIM = re.compile(r"0+p")


def _small_hex(f: float) -> str:
    return _TRIM.sub("p", float(f).hex())


def _write_mat(fh, name: str, m: Mat, vector: bool = False, row: int = 0) -> None:
    """one tensor as cformatM / cformatV print it (misc/taiyaki_flipflop5_guppy.py:38-64): a matrix one line per row of the array it was handed -- a column of
    the flappie matrix, or, for a convolution (row = padded feature count), one line per (filter, tap) --, a vector on one line"""
    fh.write("float __%s[] = {\n" % name)
    if vector:
        fh.write("\t" + ", ".join(_small_hex(x) for x in m.data.reshape(-1)))
    else:
        lines = m.data.reshape(-1, row) if row else m.data
        fh.write("\t" + ",\n\t".join(", ".join(_small_hex(x) for x in col) for col in lines))
    fh.write("};\n")
    fh.write("_Mat _%s = {\n\t.nr = %d,\n\t.nrq = %d,\n\t.nc = %d,\n\t.stride = %d,\n\t.data.f = __%s\n};\n"
             % (name, m.nr, m.nrq, m.nc, m.stride, name))
    fh.write("const flappie_matrix %s = &_%s;\n\n" % (name, name))


def tensor_names(kind: int, ident: str) -> Dict[str, str]:
    """Logical name -> symbol name, as networks.c:218-323 expects them."""
    names = {}
    if kind in (NET_LSTM5, NET_LSTM5_RLE):
        fam, cell = ("flipflop5" if kind == NET_LSTM5 else "rle5"), "lstm"
        for i in (1, 2, 3):
            names["conv%d" % i] = "conv%d_rnnrf_%s_%s_" % (i, fam, ident)
    else:
        fam, cell = "flipflop", "gru"
        names["conv1"] = "conv_rnnrf_%s_%s_" % (fam, ident)
    for i, tag in enumerate(("B1", "F2", "B3", "F4", "B5")):
        names["rnn%d" % i] = "%s%s_rnnrf_%s_%s_" % (cell, tag, fam, ident)
    names["FF"] = "FF_rnnrf_%s_%s_" % (fam, ident)
    return names


def write_mdl(path: str, model: FlipflopModel, ident: Optional[str] = None) -> None:
    """The text the reference's dump scripts print for this model, BYTE FOR BYTE (misc/taiyaki_flipflop5_guppy.py:104-164, taiyaki_rle5.py:104-164,
    taiyaki_flipflop_guppy.py:79-135) -- their quirks included: the lines their triple-quoted strings indent by four blanks (so the first array starts
    four columns in, behind the #include line's newline), a convolution's array printed one (filter, tap) per line, the GRU script's `<id>_nfilter` and
    `_conv_..._winlen` names, zero as `0x0.p+0`, no newline behind the closing #endif.  Held to the files those scripts wrote for the same numbers
    (tests/golden/ref_writer_*.mdl, tests/golden/make_mdl_fixture.py) by tests/test_mdl_reference_writer.py."""
    ident = ident or model.ident
    names = tensor_names(model.kind, ident)
    guard = "%s_%s_MODEL_H" % ("RLE" if model.kind == NET_LSTM5_RLE else "FLIPFLOP", ident.upper())
    with open(path, "w") as fh:
        fh.write("#pragma once\n    #ifndef %s\n    #define %s\n    #include \"../util.h\"\n    " % (guard, guard))
        for i, c in enumerate(model.convs):
            p = names["conv%d" % (i + 1)]
            _write_mat(fh, p + "W", c.W, row=4 * ((c.nf + 3) // 4))
            _write_mat(fh, p + "b", c.b, vector=True)
            if model.kind == NET_GRUMOD5:
                fh.write("#define %sstride  %d\n#define %s_nfilter  %d\n    #define _%swinlen  %d\n    " % (p, c.stride, ident, c.W.nc, p, c.winlen))
            else:
                fh.write("#define %sstride  %d\n#define %snfilter  %d\n    #define %swinlen  %d\n    " % (p, c.stride, p, c.W.nc, p, c.winlen))
        for i, r in enumerate(model.rnns):
            p = names["rnn%d" % i]
            _write_mat(fh, p + "iW", r.iW)
            _write_mat(fh, p + "sW", r.sW)
            _write_mat(fh, p + "b", r.b, vector=True)
        _write_mat(fh, names["FF"] + "W", model.FF_W)
        _write_mat(fh, names["FF"] + "b", model.FF_b, vector=True)
        fh.write("#endif /* %s */" % guard)


_ARR = re.compile(r"float\s+__(\w+)\[\]\s*=\s*\{(.*?)\};", re.S)
_MAT = re.compile(r"_Mat\s+_(\w+)\s*=\s*\{\s*\.nr\s*=\s*(\d+),\s*\.nrq\s*=\s*(\d+),\s*\.nc\s*=\s*(\d+),"
                  r"\s*\.stride\s*=\s*(\d+),\s*\.data\.f\s*=\s*__(\w+)\s*\};", re.S)
_DEF = re.compile(r"#define\s+(\w+?)(stride|nfilter|winlen)\s+(\d+)")


def parse_mdl_text(text: str):
    arrays = {}
    for name, body in _ARR.findall(text):
        vals = [float.fromhex(tok) for tok in body.replace("\n", " ").split(",") if tok.strip()]
        arrays[name] = np.asarray(vals, dtype=np.float32)
    mats = {}
    for name, nr, nrq, nc, stride, arr in _MAT.findall(text):
        nr, nc, stride = int(nr), int(nc), int(stride)
        data = arrays[arr]
        if data.size != nc * stride:
            raise ValueError("tensor %s: %d floats, expected nc*stride = %d" % (name, data.size, nc * stride))
        mats[name] = Mat(nr, nc, data.reshape(nc, stride).copy())
    defs = {}
    for prefix, what, val in _DEF.findall(text):
        defs[(prefix, what)] = int(val)
    return mats, defs


def load_mdl(path: str, kind: int, ident: str) -> FlipflopModel:
    with open(path) as fh:
        mats, defs = parse_mdl_text(fh.read())
    names = tensor_names(kind, ident)
    convs = []
    nconv = 1 if kind == NET_GRUMOD5 else 3
    nf = 1
    for i in range(nconv):
        p = names["conv%d" % (i + 1)]
        W, b = mats[p + "W"], mats[p + "b"]
        nf_pad = 4 * ((nf + 3) // 4)
        winlen = (W.nr - nf + nf_pad) // nf_pad
        convs.append(ConvLayer(W, b, defs[(p, "stride")], nf, winlen))
        nf = W.nc
    rnns = [RnnLayer(mats[names["rnn%d" % i] + "iW"], mats[names["rnn%d" % i] + "sW"],
                     mats[names["rnn%d" % i] + "b"]) for i in range(5)]
    return FlipflopModel(kind, convs, rnns, mats[names["FF"] + "W"], mats[names["FF"] + "b"], ident)
