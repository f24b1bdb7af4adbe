#!/bin/bash
# drop-in for the reference's sample_womancat.sh (LoRA weights, fusion_sampling_lora.py, --t_stop 0.8): same flag set; the
# hub id the reference hard-codes becomes a local diffusers-layout checkpoint folder (SD_PATH).  Without SD_PATH and
# checkpoints it runs on synthetic weights (no model files exist offline).
SD_PATH=${SD_PATH:-}
SEG_GPU=1
PROMPT="zoomed photo of a cat, lighthouse background+zoomed photo of a woman, lighthouse background+ lighthouse background"
PROMPT_ORIG="zoomed photo of a cat and a woman, lighthouse background"
RESULT_PATH="./test_out_woman"
SEED=3856
CONCEPTS="cat+woman+lighthouse"
MODIFIER="<cat1>+<woman1>+<lighthouse1>"
SEG_CONCEPTS="a cat+a woman"
PERSONAL_CHECKPOINT=${PERSONAL_CHECKPOINT:-"../checkpoint_custom/pet_cat5_lora/delta-1000.bin+../checkpoint_custom/spring_woman_lora/delta-1000.bin+../checkpoint_custom/scene_lighthouse_lora/delta-1000.bin"}
if [ -n "$SD_PATH" ]; then SRC=(--sd_path "$SD_PATH" --personal_checkpoint "$PERSONAL_CHECKPOINT"); else SRC=(--synthetic); fi
python fusion_generation/fusion_sampling_lora.py "${SRC[@]}" \
  --guidance_scale 0.8 --n_timesteps 50 --prompt "$PROMPT" \
  --output_path $RESULT_PATH --output_path_all $RESULT_PATH --sd_version "xl" --concepts "$CONCEPTS" --modifier_token $MODIFIER --resolution_h 1024 --resolution_w 1024 \
  --prompt_orig "$PROMPT_ORIG" --seed $SEED --t_cond 0.2 --t_stop 0.8 --seg_concepts="$SEG_CONCEPTS" --negative_prompt '' --seg_gpu $SEG_GPU "$@"
