run() { echo "== DEC_NB=$1 DEC_GRID=$2"; COVAHIP_DEV_DEC_NB=$1 COVAHIP_DEV_DEC_GRID=$2 python tools/quick_bench.py 256 100 2>&1 | grep "B=256\|dec" | tr "\n" " " | sed "s/x100//g; s/  */ /g"; echo; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run 0,0,0,0 0,0,0,0
run 1,0,0,0 1,0,0,0
run 1,0,0,3 1,0,0,99
run 1,0,0,4 1,0,0,99
run 1,0,0,5 1,0,0,99
run 1,0,0,6 1,0,0,99
run 1,0,0,8 1,0,0,99
run 1,0,0,12 1,0,0,99
run 1,0,3,4 1,0,99,4
run 1,0,3,6 1,0,99,4
run 1,0,4,12 1,0,99,4
run 1,2,2,12 2,4,4,4
run 0,0,0,0 0,0,0,0
