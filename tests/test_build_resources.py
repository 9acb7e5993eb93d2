"""Build-time guard (no GPU): no reachable kernel instantiation may spill registers to scratch.

The occupancy attribute on K1/K3 pins the VGPR budget (512 / waves per SIMD); a bound chosen too
high for an instantiation makes hipcc spill — K1<fp32, NCH=4, misaligned> once ran 17.6 ms instead
of 2.0 ms that way and K3<fp16, NCH=8, misaligned> 70 ms instead of 1.3 ms, with all parity tests
green.  This test reads hipcc's own resource report for the gfx950 code objects."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rnnt_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-Rpass-analysis=kernel-resource-usage", "-c"]
MAX_SPILLED_VGPRS = 8


@pytest.mark.parametrize("src", ["rnnt_mi355.hip", "joint_mi355.hip"])
def test_no_reachable_kernel_spills(src, tmp_path):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", str(tmp_path / "o.o"), os.path.join(CSRC, src)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", out.stderr)]
    assert names and len(names) == len(spills)
    bad = []
    for name, spilled in zip(names, spills):  # (only reachable instantiations exist)
        if spilled > MAX_SPILLED_VGPRS:
            bad.append((name, spilled))
    assert not bad, bad


@pytest.mark.parametrize("src", ["rnnt_mi355.hip", "joint_mi355.hip"])
def test_no_flat_memory_instructions(src, tmp_path):
    """Every access must keep its global address space.  A row pointer rebuilt from a masked integer
    (`(T*)(addr & ~15)`) compiles to flat_load/flat_store, which return out of order, so hipcc waits
    for each one before issuing the next: the misaligned-row K1 ran at 3.6 TB/s instead of 5.7 that
    way, again with all parity tests green."""
    asm = tmp_path / "k.s"
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                          "--offload-device-only", "-S", "-o", str(asm), os.path.join(CSRC, src)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    text = asm.read_text()
    assert "global_load" in text
    flat = [l.strip() for l in text.splitlines() if re.match(r"\s*flat_(load|store|atomic)", l)]
    assert not flat, flat[:5]


def _device_asm(src, tmp_path):
    asm = tmp_path / (src + ".s")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                          "--offload-device-only", "-S", "-o", str(asm), os.path.join(CSRC, src)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    text = asm.read_text()
    return {m.group(1): m.group(2) for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, flags=re.S | re.M)}


@pytest.mark.parametrize("src", ["fused_mi355.hip", "fused_bwd_mi355.hip"])
def test_mfma_kernels_stay_within_their_register_budget(src, tmp_path):
    """The MFMA kernels run two waves per SIMD at up to 256 registers.  The forward kernel sits at the limit (a few
    spilled registers outside its vocabulary walk are tolerated); the backward kernels count their LDS-DMAs by hand,
    and a scratch reload inside a step would be one more vector-memory operation on that counter: none may spill."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", str(tmp_path / "o.o"), os.path.join(CSRC, src)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", out.stderr)]
    assert names and len(names) == len(spills)
    limit = 0 if "bwd" in src else MAX_SPILLED_VGPRS
    bad = [(n, s) for n, s in zip(names, spills) if s > limit]
    assert not bad, bad


def test_fused_forward_isa_guards(tmp_path):
    """Two hazards of rnnt_fused_proj_k1 that parity tests only catch by luck:
    * the blank / label logits are re-read from global memory right after this wave stored them: the loads must be
      agent-scope (`sc1`: served by L2, never by a stale line of the CU's L1);
    * a 16-byte buffer store whose offset sits in an SGPR may be followed at once by a vector instruction that
      overwrites its data registers, and on gfx950 the store then wrote the new values: the logits stores keep their
      whole offset in the VGPR operand (soffset = 0)."""
    kernels = {k: v for k, v in _device_asm("fused_mi355.hip", tmp_path).items() if "rnnt_fused_proj_k1" in k}
    assert len(kernels) == 6       # {bf16, f16} x H in {128, 256, 512}
    for name, body in kernels.items():
        stores = re.findall(r"buffer_store_dwordx4 [^\n]*", body)
        assert stores, name
        assert all(re.search(r"s\[\d+:\d+\], 0 offen", s) for s in stores), (name, stores[:3])
        last_store = body.rindex("buffer_store_dwordx4")
        tail_loads = re.findall(r"global_load_ushort [^\n]*", body[last_store:])
        assert tail_loads and all(l.rstrip().endswith("sc1") for l in tail_loads), (name, tail_loads)


def test_fused_backward_counts_its_own_vector_memory_waits(tmp_path):
    """Inside the step loops of the backward MFMA kernels every global read is an LDS-DMA issued from inline asm and
    every `s_waitcnt vmcnt` is the kernel's own, with a hand-counted operand: a compiler-visible load there (or a
    spill reload) would add operations to the counter the waits are computed for, and a wait hipcc computes from its
    own loads would drain the DMAs in flight."""
    kernels = {k: v for k, v in _device_asm("fused_bwd_mi355.hip", tmp_path).items() if "rnnt_fused_bwd_d" in k and "merge" not in k}
    assert len(kernels) == 4       # {dH, dW} x {bf16, f16}
    for name, body in kernels.items():
        lines = body.splitlines()
        in_asm, region, seen = False, [], False
        barriers = [i for i, l in enumerate(lines) if l.strip().startswith("s_barrier")]
        assert len(barriers) >= 4, name
        for i, l in enumerate(lines):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif barriers[0] < i < barriers[-2] and not in_asm:
                region.append(t)
        foreign = [t for t in region if re.match(r"(s_waitcnt[^\n]*vmcnt|global_load|buffer_load|scratch_|flat_)", t)]
        # (vmcnt(63) is the counter's maximum: such a wait, which hipcc puts in front of a __syncthreads() it finds
        # nothing to wait for, waits for no vector-memory operation at all)
        foreign = [t for t in foreign if "vmcnt(63)" not in t]
        assert not foreign, (name, foreign[:5])
        assert "global_load_lds_dwordx4" in body and "v_mfma_f32_32x32x16" in body
