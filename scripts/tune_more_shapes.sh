#!/bin/bash
# Extends the shipped MIOpen find-db (pigeon_amd/miopen_db) with more (batch, resolution) points; run through gpurun:
#   scripts/tune_more_shapes.sh "1:1024:1024 2:1024:1024 4:1024:1024"     -> gpurun_out/miopen_db_new/
# Shapes that are not in the db fall to MIOpen's immediate-mode heuristic, which picks a 7 ms CK instance for the SDXL
# 3x3 convolutions at small batch (82 % of a batch-4 run, profiles/r01_batch4_untuned_kernel_stats_top.txt).
set -e -o pipefail
cd "$(dirname "$0")/.."
DB=$PWD/gpurun_out/miopen_db_new
[ -d $DB ] || { mkdir -p $DB && cp pigeon_amd/miopen_db/*.txt $DB/; }   # a second call in one gpurun keeps the first's finds
export MIOPEN_USER_DB_PATH=$DB
for spec in $1; do
  IFS=: read B H W <<< "$spec"
  tag=b${B}_${H}x${W}
  VAE=""; [ -n "$2" ] && VAE="--no-vae"
  timeout -k 10 150 python scripts/tune_miopen.py --collect --batch $B --height $H --width $W $VAE --shapes gpurun_out/conv_shapes_$tag.json > gpurun_out/tune_$tag.log 2>&1
  timeout -k 10 600 python scripts/tune_miopen.py --batch $B --height $H --width $W --shapes gpurun_out/conv_shapes_$tag.json >> gpurun_out/tune_$tag.log 2>&1
  echo "$tag: $(tail -1 gpurun_out/tune_$tag.log | cut -c1-60)  db lines: $(wc -l < $DB/*.ufdb.txt)"
done
