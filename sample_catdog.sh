#!/bin/bash
# same flag set as the reference's sample_catdog.sh:33-36, on synthetic weights (no checkpoints offline)
python fusion_generation/fusion_sampling.py --synthetic --seed 3821 \
  --prompt "photo of a cat running+photo of a dog running+photo of a mountain" \
  --prompt_orig "photo of a cat and a dog running, mountain background" \
  --concepts "cat+dog+mountain" --modifier_token "<new1>+<new2>+<new3>" --seg_concepts "a cat+a dog" \
  --guidance_scale 0.8 --n_timesteps 50 --t_cond 0.2 --output_path results --output_path_all results_all "$@"
