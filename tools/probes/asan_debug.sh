cd /root/repo
echo "--- step 1: import torch + cuda under asan"
tools/run_asan.sh python -X faulthandler -c "
import torch
print('avail', torch.cuda.is_available())
x = torch.zeros(4).cuda(); print('ok', x.sum().item())
" 2>&1 | grep -v "^  File" | head -40
echo "--- step 2: load the asan library, create a device handle"
tools/run_asan.sh python -X faulthandler -c "
import torch
from raw_image_pipeline_amd import RawImagePipeline
p = RawImagePipeline(False, '', '', '', device=0)
print('handle ok')
" 2>&1 | grep -v "^  File" | head -40
