"""Compile-time guard on the properties the measured performance rests on: no kernel
spills to scratch, and the tile shapes that are meant to run two workgroups per CU stay
within 128 VGPRs and 80 KB of LDS.  (hipcc's kernel-resource-usage remarks; no GPU.)"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
SRC = REPO / "larynx_amd" / "csrc" / "mi355tts.hip"


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).is_file():
        pytest.skip("hipcc not available")
    # the remarks of one source state are kept under build/ (git-ignored): the device compile takes two minutes
    import hashlib

    h = hashlib.sha1()
    for f in sorted(SRC.parent.glob("*")) + [REPO / "include" / "mi355tts.h"]:
        h.update(f.read_bytes())
    cache = REPO / "build" / f"kernel_resources_{h.hexdigest()[:12]}.txt"
    if cache.is_file():
        remarks = cache.read_text()
    else:
        out = tmp_path_factory.mktemp("res") / "dev.o"
        proc = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                               "-Rpass-analysis=kernel-resource-usage", str(SRC), "-o", str(out)],
                              capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, proc.stderr[-2000:]
        remarks = proc.stderr
        cache.parent.mkdir(exist_ok=True)
        for old in cache.parent.glob("kernel_resources_*.txt"):
            old.unlink()
        cache.write_text(remarks)
    table = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]:
        name = block.split()[0]

        def num(key):
            m = re.search(key + r": (\d+)", block)
            return int(m.group(1)) if m else -1

        # "scratch" = private memory a kernel really uses.  A kernel whose only spills are SGPRs (parked in the lanes
        # of a VGPR by v_writelane, no memory traffic) still reserves one 20-byte slot per lane for that VGPR: the
        # grouped bf16 kernels (three inlined tile bodies, 106 SGPRs) are in that position and it is not counted.
        size, vspill = num(r"ScratchSize \[bytes/lane\]"), num("VGPRs Spill")
        table[name] = dict(vgprs=num("VGPRs"), scratch=0 if (vspill == 0 and 0 < size <= 20 and num("SGPRs Spill") > 0) else size,
                           lds=num(r"LDS Size \[bytes/block\]"), occupancy=num(r"Occupancy \[waves/SIMD\]"))
    assert len(table) > 50
    return table


def conv_variants(table):
    """(K, CI_C, MB, NB, WN, KS, HALO, EPI, WM) -> resources, decoded from the mangled names."""
    out = {}
    for name, r in table.items():
        m = re.match(r"_ZN8mi355tts16conv_mfma_kernelI((?:Li\d+E){9})", name)
        if m:
            out[tuple(int(v) for v in re.findall(r"Li(\d+)E", m.group(1)))] = r
    return out


def group_variants(table):
    """conv_group_kernel<K0, K1, K2, CI_C, MB, NB, WN, KS, H0, H1, H2, WM> and the bf16 / pair group kernels."""
    out = {}
    for name, r in table.items():
        for tag, n in (("conv_group_kernel", 12), ("conv_bf16_group_kernel", 12), ("pair_group_kernel", 5), ("conv_bf16_kernel", 8)):
            m = re.match(rf"_ZN8mi355tts\d+{tag}I((?:Li\d+E){{{n}}})", name)
            if m:
                out[(tag,) + tuple(int(v) for v in re.findall(r"Li(\d+)E", m.group(1)))] = r
    return out


def test_no_used_kernel_spills(resources):
    # resblock_pair_kernel<K, 2, 2> (64 channels x 256 columns) is instantiated but never launched (it lost the sweep);
    # the 256-column NB2 tiles (MB = 2, NB = 2: batch > 1 launches only) keep a 16-20 byte spill outside their loops
    # the fused split-bf16 pair kernel at 64 channels (WM = 2, WN = 4, NB = 2) is capped at 128 VGPRs for two workgroups
    # per CU and parks 8 dwords once per tile, outside its MFMA steps
    def tolerated(n):
        return ("resblock_pair_kernelILi" in n or re.search(r"Li16ELi2ELi2ELi4ELi2E", n) or re.search(r"Li32ELi2ELi1ELi2ELi4E", n)
                or re.search(r"pair_bf16_(group_)?kernelI(Li\d+E)+?Li2ELi4ELi2ELi[13]E", n))

    spilled = {n: r["scratch"] for n, r in resources.items() if r["scratch"] > 0 and not tolerated(n)}
    assert not spilled, spilled
    fused16 = {n: r for n, r in resources.items() if "pair_bf16_group_kernel" in n}
    assert len(fused16) == 4 and all(r["scratch"] <= 48 and r["lds"] <= 80 * 1024 for r in fused16.values()), fused16
    assert all(r["vgprs"] <= 128 for n, r in fused16.items() if re.search(r"Li2ELi4ELi2ELi[13]E", n))
    pair_used = {n: r for n, r in resources.items() if re.search(r"resblock_pair_kernelILi\d+ELi(1ELi2|2ELi1)E", n)}
    assert len(pair_used) == 6 and all(r["scratch"] == 0 for r in pair_used.values())


def test_narrow_stage_kernel_fits_three_workgroups_per_cu(resources):
    mrf = {n: r for n, r in resources.items() if "mrf_small_kernel" in n}
    assert len(mrf) == 2
    for n, r in mrf.items():
        assert r["vgprs"] <= 168 and r["scratch"] == 0 and r["lds"] <= 53 * 1024 and r["occupancy"] >= 3, (n, r)
    mrf8 = {n: r for n, r in resources.items() if "mrf8_kernel" in n}  # the 8-channel stage on the 4x4x1 MFMA: 2 waves, 26 KB
    assert len(mrf8) == 1
    for n, r in mrf8.items():
        assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["lds"] <= 27 * 1024, (n, r)


def test_glow_small_launch_kernels(resources):
    """gate16 (8 waves at H = 192: <= 128 VGPRs and — round 4 — 30 KB of LDS, so that a workgroup fits the 32 KB hole a
    finishing ResBlock workgroup leaves on a loaded CU) and the 8-wave attention (its time IS its instruction count: no
    spills; the P <= 256 instantiation at 77 KB of LDS instead of the 141 KB of the P <= 768 one; the shipped voices'
    dk = 96 variant well under 128 registers)."""
    gate = {n: r for n, r in resources.items() if "gate16_kernel" in n}
    assert len(gate) == 14, sorted(gate)  # 2 tap counts x 6 widths, + the wide-pass form (two row tiles per workgroup) of H = 192
    for n, r in gate.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 128, (n, r)
    h192 = [r for n, r in gate.items() if "ILi5ELi6ELi1E" in n]
    assert len(h192) == 1 and h192[0]["lds"] <= 31 * 1024
    wide = [r for n, r in gate.items() if "ILi5ELi6ELi2E" in n]
    assert len(wide) == 1 and wide[0]["lds"] <= 32 * 1024 and wide[0]["vgprs"] <= 96


def test_round5_kernels(resources):
    """post_conv_kernel (voc_out.h): no scratch, three workgroups per CU; the attention instantiations.  (The column-owner
    WaveNet layer of round 5 is no longer in the product library: tools/probe/wn_layer.h.)"""
    assert not [n for n in resources if "wn_layer_kernel" in n]
    # lin16_kernel: two workgroups per CU for every form (<= 80 KB; the Cin = 768 FFN conv is the largest at 72 KB), and the
    # wide-pass form (four row tiles per workgroup), whose reduction scratch sets its size, at exactly 64 KB
    lin = {n: r for n, r in resources.items() if "lin16_kernel" in n}
    assert len(lin) >= 9, sorted(lin)
    for n, r in lin.items():
        assert r["scratch"] == 0 and r["lds"] <= 80 * 1024, (n, r)
    wide = [r for n, r in lin.items() if n.endswith("Lb0ELi4EEEvNS_9Lin16ArgsE")]
    assert len(wide) == 1 and wide[0]["lds"] <= 66 * 1024, sorted(lin)
    post = {n: r for n, r in resources.items() if "post_conv_kernel" in n}
    assert len(post) == 3, sorted(post)
    for n, r in post.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["lds"] <= 24 * 1024, (n, r)
    att = {n: r for n, r in resources.items() if "attention_mfma_kernel" in n}
    assert len(att) == 16, sorted(att)
    for n, r in att.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 256, (n, r)
        assert r["lds"] <= (80 if "ELi256EE" in n else 142) * 1024, (n, r)
    assert [r["vgprs"] for n, r in att.items() if "ILi48ELb1E" in n][0] <= 128


def test_continuous_stream_tile_resources(resources):
    """rb_group_kernel (rb_conv.h): four 4-wave workgroups per CU by registers (<= 128, no scratch), 32 KB of LDS each."""
    rb = {n: r for n, r in resources.items() if "rb_group_kernelILi11ELi7ELi3ELi2E" in n}  # (the 128-column variant: test_round6_late_kernels)
    assert len(rb) == 1, sorted(rb)
    for n, r in rb.items():
        assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["lds"] == 32 * 1024 and r["occupancy"] >= 4, (n, r)


def test_four_wave_pair_kernel_resources(resources):
    """rb_pair_kernel / rb_pair_group_kernel (rb_pair.h): no scratch; C = 64 at <= 168 VGPRs and 47 KB (three workgroups per
    CU), C = 32 at <= 168 VGPRs and 40 KB."""
    rbp = {n: r for n, r in resources.items() if "rb_pair_group_kernel" in n or "rb_pair_kernel" in n}
    assert len(rbp) == 8, sorted(rbp)  # 3 tap counts x 2 channel counts + 2 grouped
    for n, r in rbp.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["lds"] <= 48 * 1024 and r["occupancy"] >= 3, (n, r)


def test_two_workgroups_per_cu_where_the_schedule_counts_on_it(resources):
    conv = conv_variants(resources)
    assert conv
    groups = group_variants(resources)
    assert len(groups) >= 20
    for key, r in groups.items():
        if key[0] == "conv_group_kernel":
            K0, K1, K2, ci, MB, NB, WN, KS, h0, h1, h2, WM = key[1:]
            # the batch-1 shapes of the shipped vocoders: 64-row one-column-block tiles, the 128-column tile, the 128-row tile
            if (NB == 1 and MB == 2 and KS == 8) or (MB == 1 and NB == 2 and KS == 4) or WM == 4:
                assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["lds"] <= 80 * 1024 and r["occupancy"] >= 4, (key, r)
        if key[0] in ("conv_bf16_group_kernel", "conv_bf16_kernel"):
            assert r["scratch"] == 0 and r["occupancy"] >= 2, (key, r)  # 256-thread workgroups: >= 2 per CU; the 512-thread k-split tile: 1
    for (K, ci, MB, NB, WN, KS, halo, epi, WM), r in conv.items():
        if epi == 0 and NB == 1 and MB == 2:  # the 64-row one-column-block LINEAR tiles (stages 0/1 of the vocoder)
            assert r["vgprs"] <= 128 and r["lds"] <= 80 * 1024, ((K, ci, MB, NB, WN, KS), r)
        if NB == 2 and MB == 1 and WN == 2 and KS == 4 and epi == 0:  # the 128-column tile
            assert r["vgprs"] <= 128 and r["lds"] <= 80 * 1024, ((K, ci, MB, NB, WN, KS), r)
        if WM == 4:  # the 128-row tile
            assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["lds"] <= 80 * 1024, ((K, ci, MB, NB, WN, KS, WM), r)
    # every workgroup of 512 threads needs at least 2 waves per SIMD
    assert all(r["occupancy"] >= 2 for r in conv.values())


def test_fp16_mode_kernels(resources):
    """The native fp16 vocoder (conv_f16.h / pair_f16.h): no kernel of the mode touches scratch; the fused 128- and 64-channel
    steps fit three waves per SIMD (168 VGPRs) and three workgroups per CU of LDS, the fused narrow steps four waves and 21 KB;
    the un-fused tiles stay at two waves per SIMD (what the 256-channel stage and the upsamplers were measured on)."""
    f16 = {n: r for n, r in resources.items() if "conv_f16" in n or "pair_f16" in n or "post_f16" in n or "pack_octets" in n}
    assert len(f16) >= 25, sorted(f16)
    for n, r in f16.items():
        assert r["scratch"] == 0, (n, r)
    pairs = {n: r for n, r in f16.items() if "pair_f16_group_kernel" in n}
    assert len(pairs) == 3
    for n, r in pairs.items():
        # pair_f16_group_kernel<K0, K1, K2, MB, NB, WM, WN, ...>: WM = 2 -> the 128-row tile, MB = 1 -> the 32-row tile
        wide = "ILi11ELi7ELi3ELi2ELi2ELi2ELi2E" in n
        slim = "ILi11ELi7ELi3ELi1ELi2ELi1ELi4E" in n
        if slim:
            assert r["occupancy"] >= 4 and r["lds"] <= 21 * 1024, (n, r)
        else:
            assert r["occupancy"] == 3 and r["vgprs"] <= 168 and r["lds"] <= (40 if wide else 42) * 1024, (n, r)
    for n, r in f16.items():
        if "conv_f16" in n:
            assert r["occupancy"] >= 2 and r["lds"] <= 42 * 1024, (n, r)


def test_round6_late_kernels(resources):
    """`wn_f16_kernel` (wn_f16.h: a coupling block's WaveNet as one fp16 launch): one wave per SIMD by design — the whole register
    file of a wave is its accumulators (96) + skip sums (64) + a 10-step weight ring (120) — and NO scratch: a spilled ring slot
    would put a scratch round trip into every step; 76 KB of LDS (h + gated tile + biases) at H = 192.  `rb_group_kernel<11, 7, 3, 4>`
    (rb_conv.h, 128-column tiles of the 128-channel stage): three workgroups per CU (<= 168 VGPRs, 48 KB), no scratch."""
    wn = {n: r for n, r in resources.items() if "wn_f16_kernel" in n}
    assert len(wn) == 2, sorted(wn)
    for n, r in wn.items():
        assert r["scratch"] == 0 and r["lds"] <= 76 * 1024, (n, r)
    nb4 = {n: r for n, r in resources.items() if re.search(r"rb_group_kernelILi11ELi7ELi3ELi4E", n)}
    assert len(nb4) == 1, sorted(nb4)
    for n, r in nb4.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["occupancy"] >= 3 and r["lds"] <= 48 * 1024, (n, r)
