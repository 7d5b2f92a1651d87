cd $GRAFT_REPO_ROOT
for lib in libzipc_hip libzipc_hip_p384s5 libzipc_hip_p256s4 libzipc_hip_p192s4 libzipc_hip_p128s3; do
  echo "== $lib"
  for b in 3 4 5 6 8; do BITS=$b DATA=c2 N_STREAMS=4096 REPS=3 KERNELS=1 ZIPC_HIP_LIB=$PWD/zipc_amd/lib/$lib.so python3 tools/exp_wall.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bits $b', 'ok' if d['ok'] else 'BAD', 'ratio %.3f' % d['ratio'], 'lz_match %.3f' % d['kernels_ms']['lz_match'])"; done
  for dt in corpus text; do DATA=$dt N_STREAMS=4096 REPS=3 KERNELS=1 ZIPC_HIP_LIB=$PWD/zipc_amd/lib/$lib.so python3 tools/exp_wall.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$dt', 'ok' if d['ok'] else 'BAD', 'defl %.2f' % d['deflate_ms'], 'lz_match %.3f' % d['kernels_ms']['lz_match'])"; done
done
