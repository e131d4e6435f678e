for n in 1000 10000 30000 78000; do for r in 16 128; do
echo "N=$n rows=$r"
python tools/call_period.py --nsrc $n --rows $r --levels 0 --calls 3000 2>&1 | grep profiling
python tools/call_period.py --nsrc $n --rows $r --levels 0 --calls 3000 --opts persistent=2 2>&1 | grep profiling
done; done
