cd /tmp
EXE=$GRAFT_REPO_ROOT/oracle/_ref/speed3d_c2c
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/oracle/_ref/mpilib:$LD_LIBRARY_PATH
nproc; lscpu | grep -E "Model name|Socket|Core|Thread" 
for np in 16 32 64; do
  for n in 256 512; do
    s=$(date +%s.%N)
    timeout 170 /opt/conda/bin/mpirun -np $np $EXE stock double $n $n $n -slabs -p2p_pl 2>&1 | grep -E "Time per run|Performance|Max error" | tr '\n' ' '
    e=$(date +%s.%N)
    echo " np=$np n=$n wall=$(echo "$e - $s" | bc)"
  done
done
