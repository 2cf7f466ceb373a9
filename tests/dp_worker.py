"""One rank of the data-parallel equivalence test (tests/test_dp_equivalence_gpu.py): runs TWO train steps of a depth-4
MsVFMEncoderDecoder through the product's DP plumbing (parallel.attach: parameter broadcast, bucketed gradient all-reduce on a
side stream launched from backward, SyncBN moment / gradient exchange) and writes rank 0's results.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env (gloo: the ranks share the one GPU of the test box)
    argv: OUT.pt MODE(f32|bf16|fp16amp) [BACKEND(gloo|nccl)]   (nccl: only with WORLD_SIZE=1 and VFMSEG_DIST_SINGLE=1 on a one-GPU box)

world 1 trains on the global batch [s0, s1]; world 2 gives sample r to rank r - what DDP + SyncBatchNorm make equivalent
(configs/_base_/default_runtime.py:5, rein/models/heads/linear_head.py:44)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    os.environ["VFMSEG_DIST_BACKEND"] = backend
    import vfmseg_amd  # noqa: F401
    from tests.helpers import full_state_dict
    from vfmseg_amd import lib as L, parallel, presets
    from vfmseg_amd.optim import PEFTOptimWrapperConstructor
    from vfmseg_amd.precision import set_compute_dtype
    from vfmseg_amd.registry import MODELS
    from vfmseg_amd.segmentors import SegDataSample
    from vfmseg_amd.synth import synth_image, synth_label
    rank, world, _ = parallel.init_from_env(backend)
    torch.cuda.set_device(0)
    L.set_device_index(0)
    amp = mode == "fp16amp"   # the reference's `--amp` under DP: AmpOptimWrapper (fp16 autocast, loss scale) + gradient all-reduce
    set_compute_dtype("fp16" if amp else mode)
    depth = 4
    cfg = presets.dinov2_ms_masked(depth=depth)
    cfg["backbone"]["backbone"]["out_indices"] = [0, 1, 2, 3]
    model = MODELS.build(cfg)
    sd = full_state_dict(depth=depth)
    if rank != 0:   # DDP's constructor broadcast must make rank 0's weights win (C3): start the others from different values
        sd = {k: (v + 0.01 if v.is_floating_point() and "lora_" in k else v) for k, v in sd.items()}
    model.load_state_dict(sd)
    model = model.cuda().train()
    for m in model.modules():
        if hasattr(m, "dropout_ratio"):
            m.dropout_ratio = 0.0
        if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
            m.p = 0.0
    oc = presets.optim_cfg()
    ocw = dict(oc["optim_wrapper"], type="AmpOptimWrapper", loss_scale="dynamic") if amp else oc["optim_wrapper"]
    ow = PEFTOptimWrapperConstructor(ocw)(model, oc["param_scheduler"])
    events = []
    gs = parallel.attach(model, ow)
    if gs is not None:   # record when each bucket is launched relative to backward (overlap order)
        orig = gs.ready

        def ready(i):
            if not gs.done[i]:
                events.append(gs.buckets[i][0])
            return orig(i)
        gs.ready = ready
    keep_all = torch.rand(2, 2, 1, 32, 32, generator=torch.Generator().manual_seed(12)) > 0.2   # [step, sample]
    boxes = [(256, 768, 128, 640), (0, 512, 512, 1024)]
    logs = []
    for step in range(2):
        idx = [0, 1] if world == 1 else [rank]
        imgs = torch.cat([synth_image(1, 1024, seed=400 + 2 * step + j) for j in idx]).cuda()
        labs = torch.cat([synth_label(1, 1024, seed=400 + 2 * step + j) for j in idx])
        model.fixed_crop_box = boxes[step]
        model.aux_decoder.transformer_decoder.fixed_keep = keep_all[step][idx]
        log = model.train_step(dict(inputs=imgs, data_samples=[SegDataSample(gt_sem_seg=labs[k]) for k in range(len(idx))]), ow)
        rec = torch.tensor([float(log[k]) for k in ("decode_lr.loss_ce", "decode_hr.loss_ce", "decode_lr.acc_seg", "decode_hr.acc_seg")],
                           dtype=torch.float64)
        if world > 1:   # logged values are rank-local means: average them for the comparison
            torch.distributed.all_reduce(rec)
            rec /= world
        logs.append(rec)
    torch.cuda.synchronize()
    if amp:
        assert ow.skipped == 0 and ow.scale == 65536.0 and ow.optimizer.step_count == 2
    if rank == 0:
        state = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if "lora_" in k or not k.startswith("backbone.")}
        torch.save(dict(state=state, logs=torch.stack(logs), events=events, world=world), out_path)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
