#!/usr/bin/env python3
"""Command line of the reference (`main.py:60-177`) on the MI355X engine.

    python main.py --input_path pc_examples/mouse.npy --input_type pc_normal --out_dir out --pretrained_weights ckpt.safetensors
    python -m torch.distributed.run --nproc-per-node 8 main.py --input_dir clouds/ --input_type pc_normal ...

Same flags as the reference.  Differences, all forced by the environment (no network, no trimesh): the checkpoint is
read from `--pretrained_weights` (the reference ignores that flag and downloads `MeshAnything_350m.pth`, main.py:95-98;
`--synthetic_weights` uses the seeded random checkpoint of the tests instead); `--input_type mesh` reads .obj / .ply / .off /
.stl and samples the surface in numpy (`meshanything_amd/mesh_input.py`), `--mc` is refused (needs mesh2sdf + scikit-image); the
mesh clean-up of main.py:156-175 is restated without trimesh in `meshanything_amd/mesh_export.py`.
Multi-GPU: one process per GPU; rank r takes the shapes i % world == r and the weights travel in one RCCL broadcast.
"""
import argparse
import datetime
import os
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); keep a caller's value
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def get_args():
    p = argparse.ArgumentParser("MeshAnything", add_help=True)
    p.add_argument("--llm", default="facebook/opt-350m", type=str)
    p.add_argument("--input_dir", default=None, type=str)
    p.add_argument("--input_path", default=None, type=str)
    p.add_argument("--out_dir", default="inference_out", type=str)
    p.add_argument("--pretrained_weights", default="MeshAnything_350m.pth", type=str)
    p.add_argument("--input_type", choices=["mesh", "pc_normal"], default="pc", help="Type of the asset to process (default: pc)")
    p.add_argument("--codebook_size", default=8192, type=int)
    p.add_argument("--codebook_dim", default=1024, type=int)
    p.add_argument("--n_max_triangles", default=800, type=int)
    p.add_argument("--batchsize_per_gpu", default=1, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--mc", default=False, action="store_true")
    p.add_argument("--sampling", default=False, action="store_true")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--synthetic_weights", default=False, action="store_true", help="seeded random checkpoint (no released file offline)")
    return p.parse_args()


def main():
    from meshanything_amd import dp
    from meshanything_amd.checkpoint import load_safetensors_items, synthetic_items
    from meshanything_amd.data import Dataset
    from meshanything_amd.mesh_export import faces_from_coords, fix_normals, write_obj
    from meshanything_amd.model import MeshAnything

    args = get_args()
    rank, world, local = dp.init_process_group()
    cur_time = datetime.datetime.now().strftime("%d_%H-%M-%S")
    out_dir = os.path.join(args.out_dir, cur_time)
    os.makedirs(out_dir, exist_ok=True)

    torch.cuda.set_device(local)
    model = MeshAnything(args, device=local)
    print("load model over!!!")
    items = (lambda: synthetic_items(model.cfg)) if args.synthetic_weights else (lambda: load_safetensors_items(args.pretrained_weights))
    dp.load_weights_dp(model.engine, items, rank, world)
    print("load weights over!!!")

    if args.input_dir is not None:
        input_list = sorted(os.listdir(args.input_dir))
        if args.input_type == "pc_normal":
            input_list = [os.path.join(args.input_dir, x) for x in input_list if x.endswith(".npy")]
        else:                                    # main.py:125-128 keeps .ply / .obj / .npy for meshes; .npy is not a mesh file, .off / .stl are read too
            input_list = [os.path.join(args.input_dir, x) for x in input_list if x.lower().endswith((".ply", ".obj", ".off", ".stl"))]
    elif args.input_path is not None:
        input_list = [args.input_path]
    else:
        raise ValueError("input_dir or input_path must be provided.")
    np.random.seed(args.seed)                    # accelerate.set_seed(args.seed) before Dataset (main.py:129-133)
    torch.manual_seed(args.seed)
    dataset = Dataset(args.input_type, input_list, args.mc)

    begin = time.time()
    print("Generation Start!!!")
    mine = dp.shard_indices(len(dataset), rank, world)
    for batch in dp.batches(mine, args.batchsize_per_gpu):
        data = [dataset[i] for i in batch]
        pc = torch.from_numpy(np.stack([d["pc_normal"] for d in data])).cuda()
        outputs = model(pc, sampling=args.sampling).cpu().numpy()
        for d, coords in zip(data, outputs):
            verts, faces = faces_from_coords(coords)
            faces = fix_normals(verts, faces)
            path = os.path.join(out_dir, f'{d["uid"]}_gen.obj')
            write_obj(path, verts, faces)
            print(f"{path} Over!!")
    print(f"Total time: {time.time() - begin}")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
