for cfg in "A" "B VG_PACK_OVERLAP=1" "C VG_PACK_OVERLAP=1 VG_PACK_BPC=4" "D VG_PACK_OVERLAP=1 VG_PACK_BPC=1"; do
  set -- $cfg; name=$1; shift
  for sz in 1000000 8000000; do
    st=20; [ $sz = 8000000 ] && st=5
    env "$@" python3 bench.py --cpu-sample 0 --no-check --reads $sz --steps $st 2>/dev/null | tail -1 > gpurun_out/po_$name.json
    python3 -c "
import json,sys
j=json.load(open('gpurun_out/po_$name.json')); print('$name', $sz, '%.4g'%j['value'], '%.4f'%j['ms_per_step'], j['device_ms_per_step'])"
  done
done
