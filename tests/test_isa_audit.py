"""CPU: the hand-placed parts of csrc/attention_w4.hip are checked in the code hipcc actually emits (cross-compiled for gfx950, no GPU needed).
Its MFMAs and LDS-DMA loads are inline assembly, which hipcc neither pads for hazards nor counts (cdna_hip_programming.md 5.7): the listing must
show (1) no VALU / accvgpr write of an MFMA operand within two instructions in front of an unpadded asm MFMA, (2) no reader of an asm MFMA's
destination within 12 issue slots, (3) a main loop free of scratch traffic and accumulator-file copies."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc in this environment")
    out = tmp_path_factory.mktemp("isa") / "attention_w4.s"
    # the flags of csrc/build.sh for this file
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "videoitg_amd", "csrc", "attention_w4.hip")], check=True, capture_output=True, timeout=600)
    return str(out)


def test_no_unpadded_hazard_around_the_asm_mfmas(listing):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mfma_hazard_audit.py"), listing], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 hazard(s)" in r.stdout


def test_main_loop_has_no_scratch_and_no_accumulator_copies(listing):
    text = open(listing).read()
    kernels = re.findall(r"^(_ZN4vitg14attn_w4_kernel\w+):.*?s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) >= 2                                   # fp16 and bf16 instantiations of the shipped form
    for m in re.finditer(r"^(_ZN4vitg14attn_w4_kernel\w+):(.*?)s_endpgm", text, flags=re.S | re.M):
        body = m.group(2)
        start = body.find("Inner Loop Header")
        assert start > 0, m.group(1)
        loop = body[start:]
        # the loop body = two barrier intervals
        ends = [x.start() for x in re.finditer(r"\bs_barrier\b", loop)]
        loop = loop[:ends[1]]
        # (bounds, not exact counts: a ROCm point release may unroll or schedule differently - what matters is that the two intervals hold the tile's MFMAs)
        assert 96 <= loop.count("v_mfma_f32_32x32x16") <= 160, (m.group(1), loop.count("v_mfma_f32_32x32x16"))
        assert "scratch_" not in loop, m.group(1)
        # accumulator-file traffic only inside the (rare) rescale block, which sits behind a branch: none on the straight path between the MFMAs
        straight = loop.split("s_cbranch_scc")[0]
        assert "v_accvgpr" not in straight, m.group(1)
        assert 1 <= loop.count("s_waitcnt vmcnt(0)") <= 2, m.group(1)   # at most one drain per interval, in front of its barrier - and no other VMEM wait
