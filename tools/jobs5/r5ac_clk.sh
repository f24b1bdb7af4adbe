python tools/step_gap.py > /dev/null 2>&1 &
sleep 45
for i in 1 2 3; do rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 1500; echo; sleep 2; done
python -c "
import torch
try: print('torch clock_rate', torch.cuda.clock_rate(), 'power', torch.cuda.power_draw(), 'temp', torch.cuda.temperature(), 'util', torch.cuda.utilization())
except Exception as e: print('torch query failed', e)"
wait
