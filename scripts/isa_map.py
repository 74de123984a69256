"""Map of one kernel in a hipcc -S listing: labels, branches, barriers, sleeps, scratch (spill) traffic and VMEM, with line offsets.
    hipcc ... -S --cuda-device-only csrc/psfm_solver.hip -o /tmp/solver.s ; python scripts/isa_map.py /tmp/solver.s <mangled-prefix>"""
import sys
L = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = [i for i, l in enumerate(L) if l.startswith(pref) and ':' in l][0]
end = [i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end')][0]
body = L[start:end]
print('lines', len(body))
keys = ('scratch_', 's_barrier', 's_sleep', 's_cbranch', 's_branch', 'global_load', 'global_store', 'global_atomic', 'ds_bpermute')
quiet = len(sys.argv) > 3
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith('.LBB') or any(k in t for k in (keys[:5] if quiet else keys)):
        print(i, t[:100])
