"""Print the compiler's per-kernel resource remarks (build/kernel_resources_*.txt) for kernels whose name contains a pattern."""
import glob, re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
f = sorted(glob.glob('/root/repo/build/kernel_resources_*.txt'))[-1]
t = open(f).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)
K = {"vgpr": r"VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)"}
names = [b.split('\n')[0].strip() for b in blocks[1:]]
dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks[1:], dem):
    if pat not in d:
        continue
    vals = {k: (re.search(r, b).group(1) if re.search(r, b) else "?") for k, r in K.items()}
    d = d.replace('mi355tts::', '').replace('void ', '')
    print(f"{d[:100]:100s} " + " ".join(f"{k} {v}" for k, v in vals.items()))
