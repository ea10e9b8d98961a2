for n in ${NS:-2048 4096 8192}; do
  for spl in 0 4; do
    MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_exp.so MAGPROP_AMD_SPL=$spl python bench.py --curve --nwalk $n --no-cpu-baseline --no-extra --no-mcmc --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('curve n=$n spl=$spl', 'kernel_ms %.4f value %.3f M' % (d['roofline']['kernel_ms_avg'], d['value']/1e6), d['config'].get('kernel_variant'))"
  done
done
