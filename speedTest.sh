#!/bin/sh
# speedTest.sh <MPI-RANK> <X> <Y> <Z>   -- same CLI (plain POSIX sh: the reference is started as `sh speedTest.sh ...`)
#   as /root/reference/3dmpifft_opt/speedTest.sh:6
#   (reference: mpirun -np $1 ... ./distFFTOpt $2 $3 $4 1)
# <MPI-RANK> is the number of processes = number of GPUs (one process per GPU, RCCL over xGMI between them).  No MPI is
# needed: the processes rendezvous over TCP on localhost (dfft_boot_*, include/dfft.h).  Set DFFT_MPIRUN="mpirun ..." to
# launch through a real MPI instead (rank/size are then taken from PMI_RANK/OMPI_COMM_WORLD_RANK).
HERE=$(dirname "$(readlink -f "$0")")
BIN=${DFFT_BIN:-$HERE/distributedfft_amd/lib/distFFTOpt}
NP=${1:?usage: sh speedTest.sh <MPI-RANK> <X> <Y> <Z>}
if [ ! -x "$BIN" ]; then echo "build first: python -m distributedfft_amd.build" >&2; exit 1; fi

if [ -n "$DFFT_MPIRUN" ]; then
    exec $DFFT_MPIRUN -np "$NP" "$BIN" $2 $3 $4 1
fi

NGPU=$("$BIN" --device-count 2>/dev/null | tail -1)
NGPU=${NGPU:-0}
if [ "$NP" -gt 1 ] && [ "$NGPU" -lt "$NP" ] && [ "$NGPU" -gt 0 ] && [ "${DFFT_EXCHANGE:-}" = "ipc" ]; then
    # fewer GPUs than ranks, multi-process anyway: the hipIpc communicator lets several ranks share a GPU (round-robin)
    echo "speedTest.sh: $NGPU GPU(s) for $NP ranks -> $NP processes sharing them (DFFT_EXCHANGE=ipc)" >&2
    PORT=${DFFT_MASTER_PORT:-29533}
    pids=""
    r=0
    while [ "$r" -lt "$NP" ]; do
        DFFT_RANK=$r DFFT_WORLD_SIZE=$NP DFFT_MASTER_ADDR=127.0.0.1 DFFT_MASTER_PORT=$PORT DFFT_LOCAL_DEVICE=$((r % NGPU)) \
            DFFT_VIRTUAL_DEVICES=1 "$BIN" $2 $3 $4 1 &
        pids="$pids $!"
        r=$((r + 1))
    done
    rc=0
    for p in $pids; do wait "$p" || rc=$?; done
    exit $rc
fi
if [ "$NP" -gt 1 ] && [ "$NGPU" -lt "$NP" ]; then
    # fewer GPUs than ranks: drive <MPI-RANK> virtual devices from one process (GPU_COUNT = $NP), sharing the GPUs
    # round-robin like the reference driver's hipSetDevice(globalIdx % devCount) (fftSpeed3d_c2c.cpp:53)
    echo "speedTest.sh: $NGPU GPU(s) for $NP ranks -> one process, GPU_COUNT=$NP (virtual devices)" >&2
    DFFT_VIRTUAL_DEVICES=1 exec "$BIN" $2 $3 $4 "$NP"
fi

PORT=${DFFT_MASTER_PORT:-29533}
pids=""
r=0
while [ "$r" -lt "$NP" ]; do
    DFFT_RANK=$r DFFT_WORLD_SIZE=$NP DFFT_MASTER_ADDR=127.0.0.1 DFFT_MASTER_PORT=$PORT DFFT_LOCAL_DEVICE=$r \
        "$BIN" $2 $3 $4 1 &
    pids="$pids $!"
    r=$((r + 1))
done
rc=0
for p in $pids; do wait "$p" || rc=$?; done
exit $rc
