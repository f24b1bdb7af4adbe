#!/bin/bash
# drop-in for the reference's sample_catdog.sh (Custom-Diffusion weights, fusion_sampling.py): same flag set; the hub id
# the reference hard-codes becomes a local diffusers-layout checkpoint folder (SD_PATH).  Without SD_PATH and checkpoints
# it runs on synthetic weights (no model files exist offline).
SD_PATH=${SD_PATH:-}
SEG_GPU=1
PROMPT="photo of a cat running, mountain background+photo of a dog running, mountain background+mountain background"
PROMPT_ORIG="photo of a cat and a dog running, mountain background"
RESULT_PATH="./test_out"
SEED=3821
CONCEPTS="cat+dog+mountain"
MODIFIER="<cat1>+<dog1>+<mountain1>"
SEG_CONCEPTS="a cat+a dog"
PERSONAL_CHECKPOINT=${PERSONAL_CHECKPOINT:-"./checkpoint_custom/cat1.bin+./checkpoint_custom/dog1.bin+./checkpoint_custom/mountain1.bin"}
if [ -n "$SD_PATH" ]; then SRC=(--sd_path "$SD_PATH" --personal_checkpoint "$PERSONAL_CHECKPOINT"); else SRC=(--synthetic); fi
python fusion_generation/fusion_sampling.py "${SRC[@]}" \
  --guidance_scale 0.8 --n_timesteps 50 --prompt "$PROMPT" \
  --output_path $RESULT_PATH --output_path_all $RESULT_PATH --sd_version "xl" --concepts "$CONCEPTS" --modifier_token $MODIFIER --resolution_h 1024 --resolution_w 1024 \
  --prompt_orig "$PROMPT_ORIG" --seed $SEED --t_cond 0.2 --seg_concepts="$SEG_CONCEPTS" --negative_prompt '' --seg_gpu $SEG_GPU "$@"
