"""Per-modality time of the metric workload's encode stage (B = 48: 336 px image, 10 s audio, 8-frame video -> encoder + projector), HIP
events on the launch stream, and - under `rocprofv3 --kernel-trace --stats -- python3 tools/probes/encode_stage_profile.py` - the kernel
mix of the stage alone (the LLM is built with 1 layer and never run)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from modelcompose_amd import synthetic  # noqa: E402
from modelcompose_amd.model.builder import build_from_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    meta = bench.workload_meta("iav", 1)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    model = build_from_state_dict(meta, sd, device=dev)
    del sd
    mi = bench.synthetic_inputs(("vision", "audio", "video"), B, dev, 100)
    out = {}
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else ("vision", "audio", "video", "all")      # e.g. "audio": that modality's kernel mix alone
    for modal in only:
        inp = mi if modal == "all" else {modal: mi[modal]}
        for _ in range(2):
            model.encode_modal_inputs(inp)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            model.encode_modal_inputs(inp)
        b.record()
        torch.cuda.synchronize()
        out[modal] = round(a.elapsed_time(b) / reps, 3)
    res = {"probe": "encode_stage", "batch": B, "ms": out}
    if os.environ.get("MC_ENC_AB"):
        # in-process A/B of a library option (device-to-device spread on this pool is +-3 %: only same-process numbers compare), e.g.
        # MC_ENC_AB=tail_split: the whole stage with the option off / on, interleaved rounds
        from modelcompose_amd import _lib
        opt = os.environ["MC_ENC_AB"].encode()
        ab = {0: [], 1: []}
        for _ in range(4):
            for on in (0, 1):
                _lib.check(_lib.lib().mc_gemm_set_option(opt, on), "set_option")
                model.encode_modal_inputs(mi)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    model.encode_modal_inputs(mi)
                b.record()
                torch.cuda.synchronize()
                ab[on].append(round(a.elapsed_time(b) / reps, 3))
        _lib.check(_lib.lib().mc_gemm_set_option(opt, 0), "set_option")
        res["ab"] = {"option": os.environ["MC_ENC_AB"], "off_ms": ab[0], "on_ms": ab[1]}
    if os.environ.get("MC_ENC_STREAMS_AB"):
        # the towers on side streams (model.encode_streams) / their batches in halves or quarters (model.encode_split) against one stream,
        # interleaved rounds in this process; outputs compared bitwise
        modes = {"one_stream": (False, 1), "streams": (True, 1), "streams_split2": (True, 2), "streams_split4": (True, 4)}
        ab = {k: [] for k in modes}
        ref = None
        for _ in range(4):
            for name, (on, sp) in modes.items():
                model.encode_streams, model.encode_split = on, sp
                f, _m = model.encode_modal_inputs(mi)
                torch.cuda.synchronize()
                if ref is None:
                    ref = {k: v.clone() for k, v in f.items()}
                else:
                    assert all(torch.equal(ref[k], f[k]) for k in ref), f"{name} changed the features"
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    model.encode_modal_inputs(mi)
                b.record()
                torch.cuda.synchronize()
                ab[name].append(round(a.elapsed_time(b) / reps, 3))
        del model.encode_streams, model.encode_split
        res["encode_streams_ab"] = {"ms": ab, "bitwise_equal": True}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
