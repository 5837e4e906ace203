"""Where a kernel's scratch (spill) instructions sit relative to its MFMA blocks: python tools/spillmap.py wino_dw <mangled substring>"""
import re, subprocess, sys
tu, pat = sys.argv[1], sys.argv[2]
subprocess.run("cd /root/repo/neuralrgbd_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -S "
               "--cuda-device-only -I ../../include %s.hip -o /tmp/%s.s 2>/dev/null" % (tu, tu), shell=True, check=True)
s = open("/tmp/%s.s" % tu).read()
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if pat in m]
for name in names:
    k = s.split(name + ":")[1].split(".Lfunc_end")[0]
    blk, stats, order = "entry", {}, []
    for l in k.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk = m.group(1)
        if blk not in stats:
            stats[blk] = [0, 0, 0, 0, 0]
            order.append(blk)
        st = stats[blk]
        st[4] += 1
        st[0] += "v_mfma" in l
        st[1] += "scratch_load" in l
        st[2] += "scratch_store" in l
        st[3] += "s_barrier" in l
    print(name)
    for b in order:
        st = stats[b]
        if st[0] or st[1] or st[2]:
            print("  %-12s lines %5d mfma %3d scratch_load %3d scratch_store %3d barrier %d" % (b, st[4], st[0], st[1], st[2], st[3]))
