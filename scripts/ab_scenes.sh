#!/bin/bash
# scripts/ab_scenes.sh "<variant names>" -- scripts/ab_variants.sh over the headline and the three big scenes (inside gpurun)
cd "$GRAFT_REPO_ROOT" || exit 1
names=$1
echo "== headline"; bash scripts/ab_variants.sh "$names $names"
echo "== terrain"; bash scripts/ab_variants.sh "$names $names" --scene terrain --width 1024 --height 1024 --spp 32
echo "== C4 (32 spp)"; bash scripts/ab_variants.sh "$names $names" --scene material-ball --width 1920 --height 1080 --spp 32
echo "== C5 (16 spp)"; bash scripts/ab_variants.sh "$names $names" --scene instanced --width 2048 --height 2048 --spp 16
