"""Build-time guard for the fused forward: its 12-wave workgroups run on a 168-register budget and the compiler has answered small
source changes with 40-300 spilled registers in the ring loop before.  The headline instantiation must keep its
spills where they are today - a handful, all in phase 1 / the fallback path - and every instantiation must fit the budget."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_fused_forward_register_budget(tmp_path):
    # the kernel is compiled as three translation units (corr_fused.hip: even K at C = 384 / 768; _odd: odd K; _c192: C = 192)
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(name):
        src = os.path.join(ROOT, "stego_amd", "csrc", name)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "stego_amd", "csrc"),
               "-I", os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / (name + ".o")), "-Rpass-analysis=kernel-resource-usage"]
        return subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))

    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(compile_one, ["corr_fused.hip", "corr_fused_odd.hip", "corr_fused_c192.hip", "corr_fused_half.hip"]))
    kernels = {}
    per_unit = []
    for res in results:
        assert res.returncode == 0, res.stderr[-2000:]
        name = None
        before = len(kernels)
        for line in res.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                kernels[name] = {}
                continue
            m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
            if m and name:
                kernels[name][m.group(1)] = int(m.group(2))
        per_unit.append(len([k for k in list(kernels)[before:] if "corr_fused_kernel" in k]))
    assert per_unit == [16, 16, 12, 0], per_unit                # even K: 2 precisions x 2 widths x 4 code-chunk counts; odd K: the same; C = 192: 2 x 3 x even / odd
    # round 6: the column-half kernel of small batches (corr_fused_half.hip): the same 12-wave / 168-register budget, spills only in phase 1;
    # its sixteen-wave form (C = 384: eight MFMA waves, one row pair per wave in phase 1) lives on 128 registers, four waves per SIMD
    half = {k: v for k, v in kernels.items() if "corr_fused_half_kernel" in k}
    assert len(half) == 24, sorted(kernels)                     # 2 precisions x (2 widths x 12 waves + C = 384 x 16 waves) x 4 code-chunk counts, even K
    for k, v in half.items():
        if k.endswith("Li16EEEvNS_11FusedParamsE"):
            assert v["VGPRs"] + v.get("AGPRs", 0) <= 128 and v["Occupancy [waves/SIMD]"] >= 4, (k, v)
            assert v["VGPRs Spill"] <= 48 and v["ScratchSize [bytes/lane]"] <= 192, (k, v)       # (phase 1's taps; none between the ring loop's barriers)
        else:
            assert v["VGPRs"] + v.get("AGPRs", 0) <= 168 and v["Occupancy [waves/SIMD]"] >= 3, (k, v)
            assert v["VGPRs Spill"] <= 32 and v["ScratchSize [bytes/lane]"] <= 128, (k, v)
    fused = {k: v for k, v in kernels.items() if "corr_fused_kernel" in k}
    assert len(fused) == 44, sorted(kernels)                    # 2 precisions x (2 widths x 4 + C = 192 x 3) code-chunk counts x even / odd K
    for k, v in fused.items():
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 168, (k, v)    # 12 waves per workgroup = 3 per SIMD
        assert v["Occupancy [waves/SIMD]"] >= 3, (k, v)
    head = fused["_ZN5stego17corr_fused_kernelILi1ELi3ELi3ELb0EEEvNS_11FusedParamsE"]  # f16x3, C = 384, K <= 96, even: BASELINE config 2
    assert head["VGPRs Spill"] <= 24 and head["ScratchSize [bytes/lane]"] <= 96, head


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_backbone_gemm_register_budget(tmp_path):
    """The ViT GEMM runs two 512-thread workgroups per CU on 128 registers per lane; its F16X3 k loop holds 96 accumulators + 40 operand
    registers.  An epilogue that finished all 96 values of a wave before its passes made the compiler spill 400 registers INSIDE the k
    loop (42 ms instead of 14 for the forward): the spill counts of today are pinned."""
    src = os.path.join(ROOT, "stego_amd", "csrc", "vit_forward.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "stego_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / "vit.o"), "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    kernels, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1)] = int(m.group(2))
    gemm = {k: v for k, v in kernels.items() if "vit_gemm_kernel" in k}
    assert len(gemm) == 8, sorted(kernels)                       # 4 epilogues x 2 precisions
    for k, v in gemm.items():
        assert v["Occupancy [waves/SIMD]"] >= 4, (k, v)          # two workgroups per CU
        qkv = "ILi3E" in k                                       # (the QKV epilogue has always spilled a few dozen registers: outside the k loop)
        assert v["VGPRs Spill"] <= (48 if qkv else 8), (k, v)
    attn = {k: v for k, v in kernels.items() if "vit_attn_kernel" in k}
    assert len(attn) == 2
    for k, v in attn.items():
        assert v["VGPRs Spill"] == 0 and v["Occupancy [waves/SIMD]"] >= 2, (k, v)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_backward_kernels_do_not_spill(tmp_path):
    """Every kernel of the loss backward (tiles with and without builders in both precisions, the list and the row unsample) keeps its
    working set in registers today; the training pair - corr_bwd_tile_build_kernel<5> (K = 70) and corr_unsample_list_kernel<1, 1> - at the
    occupancies the workgroup sizes of the launch need."""
    src = os.path.join(ROOT, "stego_amd", "csrc", "corr_bwd.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "stego_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / "bwd.o"), "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    kernels, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1)] = int(m.group(2))
    assert len(kernels) >= 40, sorted(kernels)
    for k, v in kernels.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
    assert kernels["_ZN5stego26corr_bwd_tile_build_kernelILi5EEEvNS_9BwdParamsE"]["Occupancy [waves/SIMD]"] >= 2        # 512 threads: one workgroup per CU
    assert kernels["_ZN5stego25corr_unsample_list_kernelILi1ELi1EEEvNS_9BwdParamsE"]["Occupancy [waves/SIMD]"] >= 2     # 256 threads: two per CU


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_wide_path_kernels_spill_nothing(tmp_path):
    """feature_samples 12 .. 16 (csrc/corr_wide.hip, csrc/dense_corr.hip): the eight-wave kernels run on 256 registers per lane - the backward
    GEMM kernel holds 96 accumulators + 32 prefetched values, the row-pair correlation kernel the A fragments of two chunks; a scratch reload in
    either loop drains the prefetch / the ring (an unroll of 2 in wide_bwd_kernel spilled 17 registers; the row-pair kernel at six chunks 62)."""
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(name):
        src = os.path.join(ROOT, "stego_amd", "csrc", name)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "stego_amd", "csrc"),
               "-I", os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / (name + ".o")), "-Rpass-analysis=kernel-resource-usage"]
        return subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))

    with ThreadPoolExecutor(3) as ex:
        results = list(ex.map(compile_one, ["corr_wide.hip", "dense_corr.hip", "sample_sets.hip"]))
    kernels = {}
    for res in results:
        assert res.returncode == 0, res.stderr[-2000:]
        name = None
        for line in res.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                kernels[name] = {}
                continue
            m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
            if m and name:
                kernels[name][m.group(1)] = int(m.group(2))
    want = ["wide_bwd_kernel", "wide_code_tiles_kernel", "wide_pointwise_kernel", "wide_rows_kernel", "wide_gather_kernel", "dense_rowpair_kernel",
            "dense_rowblock_kernelILb1ELi2", "sample_panels_kernel", "sample_scatter_kernel"]
    for w in want:
        hits = {k: v for k, v in kernels.items() if w in k}
        assert hits, (w, sorted(kernels))
        for k, v in hits.items():
            assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (k, v)
    bwd = [v for k, v in kernels.items() if "wide_bwd_kernel" in k][0]
    assert bwd["Occupancy [waves/SIMD]"] >= 2, bwd               # eight waves per workgroup = two per SIMD


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_dense_stream_kernel_does_not_spill(tmp_path):
    """dense_stream_kernel runs one 512-register wave per SIMD with 192 registers of A fragments, 64 accumulators, one fragment set and one
    raw set; two fragment sets, row scales found on the fly or a second raw set all made the compiler spill INSIDE the chunk loop, where
    every scratch reload drains the prefetch (137 us instead of 87, profiles/r06_dense_attempts.txt).  Today: no spills but ten registers of
    the six-chunk form, reloaded at the block boundaries (none between the barriers of the steady-state chunks); pinned."""
    src = os.path.join(ROOT, "stego_amd", "csrc", "dense_stream.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "stego_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / "ds.o"), "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    kernels, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1)] = int(m.group(2))
    stream = {k: v for k, v in kernels.items() if "dense_stream_kernel" in k}
    assert len(stream) == 5, sorted(kernels)                      # 2 .. 6 chunks of 64 channels
    for k, v in stream.items():
        six = k.endswith("ILi6EEEvNS_17DenseStreamParamsE")
        assert v["VGPRs Spill"] <= (12 if six else 0) and v["ScratchSize [bytes/lane]"] <= (64 if six else 0), (k, v)
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 512, (k, v)
