for v in ${SPLITS:-14336 20480 25600 30720}; do
  echo "== $v"
  for w in l6_lowent_huff l9_lowent l1_text; do
    ZAMD_INF_SPLIT_SEGMENTS=$v timeout -k 10 200 python bench.py --workload $w --chunks-per-gpu 4096 --no-cpu-baseline --no-serial-pass 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); i=d['inflate']; print('$w', i['value'], i['decode_ms'], i['resolve_ms'])"
  done
done
