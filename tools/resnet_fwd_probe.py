#!/usr/bin/env python
"""The ResNet forward of the metric batch (64 uint8 frames 224 px) on its own, a few times -- the target of PMC passes that ask WHY a
kernel takes what it takes (tools/pmc_kernel_table.py turns the counter collection into one row per kernel):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \\
              --kernel-trace --output-format csv -d out/sq -o sq -- python tools/resnet_fwd_probe.py
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d out/tcc -o tcc -- python tools/resnet_fwd_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from clipbert_amd import modeling as M  # noqa: E402
from clipbert_amd import synthetic as S  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = dict(bench.BASE_CONFIG)
    model = M.ClipBert(cfg, detectron2_model_cfg="R-50-grid.yaml", transformer_cls=M.ClipBertForVideoTextRetrieval)
    model.load_state_dict(S.full_state_dict(cfg, "retrieval", 42), strict=True)
    model.to(dev).eval()
    model.prepare(dtype=torch.bfloat16, device=dev)
    frames = S.synthetic_frames(32, 2, 224, 42).to(dev)            # 32 clips x 2 frames = the 64 frames of the metric step
    with torch.no_grad():
        for _ in range(3):
            model.grid_features(frames)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
