export PYTHONPATH=.
mkdir -p gpurun_out/r4h
for loop in pseudo_lr knn_prop2; do
python tools/profile_loop.py $loop 1109 > gpurun_out/r4h/cum_$loop.log 2>&1
done
