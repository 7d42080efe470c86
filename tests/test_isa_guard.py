"""The library must not contain the packed-fp32 form that fails beside another wave's 16-bit MFMAs on MI355X
(v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]: wrong low half in lanes 48-63; DESIGN.md section 5.4, profiles/r02_pk_probe.txt).
tools/check_isa.py disassembles the gfx950 code objects of flappie_amd/libffhip.so; the build runs it too."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _tool():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(HERE, "..", "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_kernel_has_the_failing_packed_form():
    lib = os.path.join(HERE, "..", "flappie_amd", "libffhip.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    bad, nkern, ninst = _tool().scan(lib)
    assert nkern > 100 and ninst > 100000            # the disassembly really covers the device code
    assert not bad, bad[:5]


def test_the_guard_sees_the_form_where_it_is_issued_on_purpose():
    # the probe kernel (ffhip_debug_pk_probe) issues every op_sel form: 4 + 4 + 4 of them match, in each of its 3 instantiations
    lib = os.path.join(HERE, "..", "flappie_amd", "libffhip.so")
    bad, _, _ = _tool().scan(lib, exempt=())
    assert len(bad) == 36 and all("k_pk_probe" in k for k, _ in bad), (len(bad), bad[:3])
