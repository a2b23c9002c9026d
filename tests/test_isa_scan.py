"""The compiled device code keeps the properties the round-4 kernel work established (tools/isa_scan.py): a static check of
libxm3d_hip.so's gfx950 code objects on the CPU - no GPU needed, hipcc cross-compiles and llvm-objdump disassembles here."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import isa_scan

    if not os.path.exists(isa_scan.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    so = os.path.join(ROOT, "xmask3d_amd", "libxm3d_hip.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    r = {isa_scan.short(k): c for k, c in isa_scan.scan(so).items() if "xm3d" in k}
    assert len(r) > 200, "the scan found too few kernels: bundle format changed?"
    return r


def test_attention_forward_keeps_its_scores_in_vgprs(kernels):
    """attention.hip / attention_f32.hip are built with the MFMA destination in VGPRs: the online softmax reads every score, and AGPR
    accumulators cost a v_accvgpr_read (+ write on the masked path) per score and a third of the occupancy"""
    fwd = {k: c for k, c in kernels.items() if re.match(r"k_attn_fwd(_f32acc)?<", k)}
    assert len(fwd) >= 20
    for k, c in fwd.items():
        dq = int(re.search(r"<(\d+),", k).group(1))
        assert c["mfma"] > 0, k
        if dq <= 128:  # (the 160-channel instantiations overflow into a few AGPRs; nothing per score)
            assert c["accvgpr"] == 0, (k, dict(c))
        assert c["scratch"] == 0, (k, dict(c))


def test_no_run_time_indexed_register_arrays_in_the_gemm_epilogues(kernels):
    """`h ? acc[i] : acc[8 + i]` was lowered to a 16-way compare / select cascade per element (945 v_cmp_eq_u32 in the GEGLU kernels); the
    implicit-GEMM instantiations keep their ~80 tap-validity compares"""
    gemm = {k: c for k, c in kernels.items() if k.startswith("k_gemm<")}
    assert len(gemm) >= 40
    for k, c in gemm.items():
        assert c["cmp_eq"] <= 100, (k, dict(c))
        assert c["scratch"] <= 8, (k, dict(c))  # at most a spill store + reload pair around the K loop


def test_inference_kernels_do_not_spill(kernels):
    """scratch traffic only in the 160-channel attention backward (512 registers, training in bf16) and the GEMM's loop-invariant pair"""
    bad = {k: c["scratch"] for k, c in kernels.items() if c["scratch"] > 8 and not k.startswith("k_attn_bwd")}
    assert not bad, bad


def test_split_pass_reads_its_groupnorm_table_as_vectors(kernels):
    """k_split_nhwc was vector-memory-INSTRUCTION bound (2.2 TB/s) while it read the per-channel scale / shift table with eight 4-byte loads
    per 4-channel group; with two 16-byte loads it streams at 5.9 TB/s (profiles/r04_split_pass_bench.log)"""
    c = kernels["k_split_nhwc"]
    assert c["narrow_loads"] == 0 and c["wide_loads"] >= 4, dict(c)
