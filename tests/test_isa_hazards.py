"""Build-side check (no GPU): the kernels that issue LDS reads from inline asm and wait for them later (transpose reads,
ds_read_b128 under counted lgkmcnt) are compiled to gfx950 assembly and walked with the hardware's in-order LDS queue.
The compiler takes an asm-issued read as complete, so it is free to copy its registers before the asm wait: tools/
check_lds_hazards.py flags any vector instruction that reads such a register while the read is still outstanding (the
round-3 accumulation-pass bug was exactly that: a v_mov of transpose-read results ahead of the s_waitcnt)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))

FILES = ["aggregate_pipe.hip", "aggregate_rel.hip", "aggregate_relg.hip", "aggregate.hip", "attention_lds.hip", "attention_train.hip",
         "linear_planes.hip", "grouped.hip", "navfuse.hip"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("name", FILES)
def test_no_read_of_an_outstanding_lds_result(name):
    import check_lds_hazards as C
    found = C.check_file(os.path.join(C.CSRC, name))
    assert not found, found[:6]


def test_checker_sees_a_planted_hazard():
    import check_lds_hazards as C
    asm = """_Zk:                                    ; @_Zk
\tds_read_b64_tr_b16 v[10:11], v75
\tds_read_b64_tr_b16 v[12:13], v76
\tv_mov_b64_e32 v[16:17], v[12:13]
\ts_waitcnt lgkmcnt(0)
\tv_mov_b64_e32 v[18:19], v[10:11]
\tds_read_b128 v[20:23], v1
\tds_read_b128 v[24:27], v1 offset:16
\ts_waitcnt lgkmcnt(1)
\tv_add_f32_e32 v2, v20, v21
\tv_add_f32_e32 v3, v24, v25
.Lfunc_end0:
"""
    found = C.hazards(asm)
    assert [t for _, t in found] == ["v_mov_b64_e32 v[16:17], v[12:13]", "v_add_f32_e32 v3, v24, v25"]
