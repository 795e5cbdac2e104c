for sr in 0 4 6 12 16; do
echo "== sample.srows=$sr, one frame per call pair"
python bench.py --frames-per-call 1 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --opt sample.srows=$sr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['verified']['ok'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})"
done
