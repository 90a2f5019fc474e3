"""VGPR / scratch use per kernel from a device-only assembly dump:
   hipcc --offload-arch=gfx950 <flags of __graft_entry__> --cuda-device-only -S -o /tmp/k.s rt_amd/csrc/rt_kernels.hip
   python tools/vgprs.py /tmp/k.s [name filter ...]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2:] or ["k_walk", "k_shade", "packet"]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if not any(w in name for w in want):
        continue
    vgpr = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)
    acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
    scratch = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)
    print(f"{vgpr:>4} vgpr  accum_offset {acc.group(1) if acc else '-':>4}  scratch {scratch:>5}  {name[:110]}")
