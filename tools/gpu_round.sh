#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
for cfg in "16 16 16" "32 16 16" "32 16 32" "32 8 32" "24 8 16"; do
set -- $cfg
HS_ED25519_POOL_NARROW=$1 HS_ED25519_POOL_PARTITIONS=$2 GPU_MAX_HW_QUEUES=$3 python - <<'PY'
import ctypes, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import synth
from hotstuff_amd import _lib
_lib.init(0,1024)
lib=_lib.lib()
n=1024
bt=[synth.make_batch(n, shared=False, first=j*n, cache_dir='/tmp/hs_synth') for j in range(16)]
dg=b"".join(b[0] for b in bt); pk=b"".join(b[1] for b in bt); sg=b"".join(b[2] for b in bt)
bufs=[ctypes.create_string_buffer(x,len(x)) for x in (dg,pk,sg)]
a=[ctypes.addressof(x) for x in bufs]
out=[]
for th in (16,24,32,48):
    calls,p50,el=ctypes.c_uint64(),ctypes.c_double(),ctypes.c_double()
    rc=lib.hs_ed25519_loadgen(a[0],32,a[1],a[2],n,16,1,th,0.5,ctypes.byref(calls),ctypes.byref(p50),ctypes.byref(el))
    out.append('%d thr: %.1f M/s p50 %.3f' % (th, calls.value*n/el.value/1e6, p50.value))
print('narrow', os.environ['HS_ED25519_POOL_NARROW'], 'partitions', os.environ['HS_ED25519_POOL_PARTITIONS'], 'hwq', os.environ['GPU_MAX_HW_QUEUES'], '|', ' | '.join(out))
PY
done
