# sweep of the phase plan of the certified batch pass (mvdb.hip launch_half_pass): growth of the earlier phases / of the last one
for rows in 10000000 1000000; do
for nq in 256 128 32; do
for pg in "16 6" "6 4" "4 4"; do
set -- $pg
MVDB_HALF_PHASE_GROWTH=$1 MVDB_HALF_LAST_GROWTH=$2 timeout 300 python3 bench.py --rows $rows --nq $nq --dim ${DIM:-512} --steps 60 --warmup 10 --no-cpu-baseline --no-encoder 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d.get('power') or {}; print('rows=$rows nq=$nq growth=$1/$2', d['value'], d['ms_per_step'], r['launches'], r['avg_launch_ms'], p.get('package_w'), p.get('sclk_mhz'))"
done; done; done
