#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pp in 16 8 4; do
HS_ED25519_POOL_PARTITIONS=$pp python - <<'PY'
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
for th in (8,16,24,32):
    calls,p50,el=ctypes.c_uint64(),ctypes.c_double(),ctypes.c_double()
    rc=lib.hs_ed25519_loadgen(a[0],32,a[1],a[2],n,16,1,th,0.5,ctypes.byref(calls),ctypes.byref(p50),ctypes.byref(el))
    print(os.environ['HS_ED25519_POOL_PARTITIONS'],'threads',th,'rc',rc,'M/s %.1f'%(calls.value*n/el.value/1e6),'p50 %.3f'%p50.value)
PY
done
