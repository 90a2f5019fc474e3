#!/bin/bash
# Copies the outputs of tools/prof_round.sh (gpurun_out/stats_<tag>_{default,pipes1}, gpurun_out/prof_<tag>) into profiles/
# under the round's names.  usage: tools/install_profiles.sh <tag> <profile prefix, e.g. r03_h> <workload name, e.g. heightfield1m>
set -eu
TAG=$1; PRE=$2; NAME=$3
cp "$(find gpurun_out/stats_${TAG}_default/trace -name '*kernel_stats.csv' | head -1)" profiles/${PRE}_kernel_stats_${NAME}.csv
cp "$(find gpurun_out/stats_${TAG}_pipes1/trace -name '*kernel_stats.csv' | head -1)" profiles/${PRE}_kernel_stats_${NAME}_pipes1.csv
cp gpurun_out/prof_${TAG}/summary.txt profiles/${PRE}_pmc_${NAME}.txt
cp gpurun_out/prof_${TAG}/traffic.json profiles/${PRE}_traffic_${NAME}.json
echo "installed ${PRE}_*_${NAME}*"
