"""Packed-fp32 instructions beside another kernel's MFMAs: which op_sel forms return wrong halves, in which lanes?
(ffhip_debug_pk_probe).  The probe runs on its own stream while a batch's recurrent layers run."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M, binding as B

eng = B.Engine(0)
lib = B.lib()
lib.ffhip_debug_pk_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint)]
OPS = ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_mov_b32")

def probe(iters, nwg, ballast):
    buf = (C.c_uint * (4 * 16 * 2 * 4))()
    rc = lib.ffhip_debug_pk_probe(eng.h, iters, nwg, ballast, buf)
    assert rc == 0, rc
    return np.frombuffer(buf, dtype=np.uint32).reshape(4, 16, 2, 4).copy()

def report(tag, c):
    tot = int(c.sum())
    print("%-70s mismatches: %d" % (tag, tot))
    if tot:
        for op in range(4):
            for f in range(16):
                if c[op, f].sum():
                    if op == 3 and f in (1, 2):
                        print("     v_pk_fma_f32 op_sel:[%d,%d,1] op_sel_hi:[1,1,0]   low half by wave quarter %s   high half %s" % (f - 1, f - 1, c[op, f, 0].tolist(), c[op, f, 1].tolist()))
                        continue
                    print("     %s op_sel:[%d,%d] op_sel_hi:[%d,%d]   low half by wave quarter %s   high half %s" % (
                        OPS[op], f >> 3 & 1, f >> 2 & 1, f >> 1 & 1, f & 1, c[op, f, 0].tolist(), c[op, f, 1].tolist()))

iters, nwg = int(os.environ.get("ITERS", "20000")), int(os.environ.get("NWG", "2048"))
for ballast in (1, 96):
    report("probe alone, ballast %d" % ballast, probe(iters, nwg, ballast))
rng = np.random.default_rng(5)
cases = [("LSTM H=256, 256 reads (one tile per group, 2 workgroups per CU)", M.NET_LSTM5, 256, 256, 0),
         ("LSTM H=256, 512 reads (pair form, 2 workgroups per CU)", M.NET_LSTM5, 256, 512, 0),
         ("GRUmod H=256, 256 reads", M.NET_GRUMOD5, 256, 256, 0),
         ("LSTM H=384, 256 reads (pair form, 1 workgroup per CU)", M.NET_LSTM5, 384, 256, 0),
         ("LSTM H=512, 256 reads", M.NET_LSTM5, 512, 256, 0),
         ("LSTM H=128, 256 reads", M.NET_LSTM5, 128, 256, 0),
         ("LSTM H=256, 256 reads, f32-MFMA persistent kernels", M.NET_LSTM5, 256, 256, B.RUN_F32_RNN),
         ("LSTM H=384, 256 reads, f32-MFMA persistent kernels", M.NET_LSTM5, 384, 256, B.RUN_F32_RNN)]
T = int(os.environ.get("T", "30000"))
for name, kind, hidden, nread, flags in cases:
    mdl = M.synthetic_model(kind, hidden, seed=3)
    dm = B.DeviceModel(eng, mdl)
    b = B.Batch(dm, nread, T)
    b.set_signals(rng.standard_normal((nread, T)).astype(np.float32))
    for ballast in (1, 96):
        b.run(1.0, flags | B.RUN_NO_DECODE)
        c = probe(iters, nwg, ballast)
        b.finish()
        report("beside %s, ballast %d" % (name, ballast), c)
    b.close(); dm.close()
