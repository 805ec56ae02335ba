#!/usr/bin/env python3
"""Vector-ALU instructions per MFMA in a kernel's hottest loop, from the device assembly (on gfx950 every vector
instruction of an f32-MFMA kernel is paid in matrix time, so this count is the first thing to look at).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o k.s speak-hack_amd/csrc/conv_inst_3x3s1_b.hip
    grep -n "^_Z.*:" k.s                       # first / last line of the kernel of interest
    python tools/isa_loop_stats.py k.s FIRST LAST

Prints the loop (backward branch target .. branch) with the most MFMAs: instruction, MFMA, VALU, LDS, VMEM and scalar counts and
the VALU histogram.  (Multi-block loops: blocks outside the span of the innermost backward branch are not counted.)"""
import sys,re
lines=open(sys.argv[1]).read().split('\n')
start=int(sys.argv[2]); end=int(sys.argv[3])
body=lines[start-1:end-1]
# find the inner loop with most mfma
loops=[]; cur=None
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
best=None
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        a=labels[m.group(1)]; seg=body[a:i+1]
        nm=sum('v_mfma' in x for x in seg)
        if best is None or nm>best[0]: best=(nm,a,i,seg)
nm,a,i,seg=best
ins=[x.strip() for x in seg if x.startswith('\t') and not x.strip().startswith(';') and not x.strip().startswith('.')]
valu=[x for x in ins if x.startswith('v_') and 'mfma' not in x]
from collections import Counter
print(f"loop lines {a}-{i}: {len(ins)} instrs, {nm} MFMA, {len(valu)} VALU, {sum(x.startswith('ds_') for x in ins)} LDS, {sum(x.startswith('global_') or x.startswith('buffer_') for x in ins)} VMEM, {sum(x.startswith('s_') for x in ins)} SALU/s_*")
print(Counter(x.split()[0] for x in valu).most_common(14))
