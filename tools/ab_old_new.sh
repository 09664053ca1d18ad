R=$GRAFT_REPO_ROOT
for i in 1 2; do
  GPFLOWSLIM_HIP_LIB=$R/gpflow-slim_amd/lib_old/libgpflowslim_hip.so timeout -k 10 200 python tools/configs.py 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('old', j['conditional_white_True_ms'], j['conditional_white_False_ms'], j['conditional_white_True_ms_plain_leaves'], j['svgp_elbo_full_q_sqrt_whitened_ms'])"
  timeout -k 10 200 python tools/configs.py 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('new', j['conditional_white_True_ms'], j['conditional_white_False_ms'], j['conditional_white_True_ms_plain_leaves'], j['svgp_elbo_full_q_sqrt_whitened_ms'])"
done
for i in 1 2; do
  GPFLOWSLIM_HIP_LIB=$R/gpflow-slim_amd/lib_old/libgpflowslim_hip.so timeout -k 10 200 python tools/one_eval.py 32768 5 2>/dev/null | tail -1 | cut -c1-200
  timeout -k 10 200 python tools/one_eval.py 32768 5 2>/dev/null | tail -1 | cut -c1-200
done
