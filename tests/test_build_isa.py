"""The shipped library must not contain the packed-fp32 instruction form that MI355X executes wrongly beside MFMA waves of another
workgroup (alignn_amd/build.py, DESIGN.md section 4.6, tools/pk_f32_repro2.hip): v_pk_{fma,mul,add}_f32 with op_sel taking the
HIGH half of src1 for the LOW result.  The check disassembles the gfx950 code objects of the linked file - no GPU needed."""
import os

import pytest

from alignn_amd import build as B


@pytest.mark.skipif(not os.path.exists(B._OBJDUMP), reason="llvm-objdump of the ROCm image not found")
def test_no_faulting_packed_fp32_form_in_the_library():
    lib = B.build()
    assert B.faulting_packed_forms(lib) == []


def test_the_scanner_recognises_the_form():
    pk, sel = B._PK, B._OPSEL
    bad = "\tv_pk_fma_f32 v[34:35], v[34:35], v[42:43], v[36:37] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]"
    ok = ["\tv_pk_fma_f32 v[0:1], v[2:3], v[0:1], v[4:5] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]",
          "\tv_pk_mul_f32 v[32:33], v[30:31], v[32:33] op_sel_hi:[0,1]", "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]",
          "\tv_pk_fma_f32 v[0:1], v[2:3], v[0:1], v[4:5] op_sel:[1,0,0]", "\tv_pk_fma_f32 v[0:1], v[2:3], v[0:1], v[4:5] op_sel:[0,0,1]"]
    m = pk.search(bad)
    assert m and sel.search(m.group(2)).group(2) == "1"
    for line in ok:
        m = pk.search(line)
        o = sel.search(m.group(2))
        assert o is None or o.group(2) == "0", line
    assert pk.search("\tv_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]") is None
