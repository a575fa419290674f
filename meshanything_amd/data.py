"""Input side of the hot path: the reference's `Dataset` (main.py:15-58) for point-cloud and mesh-file inputs."""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np


def normalize_pc(pc_normal: np.ndarray) -> np.ndarray:
    """Dataset.__getitem__ (main.py:45-58): centre xyz on the bounding-box mid-point, scale so that max |coord| = 0.9995,
    keep the normals, return float16.  All arithmetic stays in the input's dtype (an fp16 .npy is normalised in fp16),
    exactly as numpy does it in the reference."""
    pc_coor = pc_normal[:, :3]
    normals = pc_normal[:, 3:]
    bounds = np.array([pc_coor.min(axis=0), pc_coor.max(axis=0)])
    pc_coor = pc_coor - (bounds[0] + bounds[1])[None, :] / 2
    pc_coor = pc_coor / np.abs(pc_coor).max() * 0.9995
    assert (np.linalg.norm(normals, axis=-1) > 0.99).all(), "normals should be unit vectors, something wrong"
    return np.concatenate([pc_coor, normals], axis=-1, dtype=np.float16)


class Dataset:
    """`Dataset('pc_normal' | 'mesh', paths)` of main.py:15-58.  Sampling uses the GLOBAL numpy RNG like the reference
    (seed it first: main.py:129-133 calls accelerate.set_seed(args.seed) -> np.random.seed)."""

    def __init__(self, input_type: str, input_list: List[str], mc: bool = False, n_points: int = 4096):
        self.data: List[Dict] = []
        if input_type == "pc_normal":
            for input_path in input_list:
                cur_data = np.load(input_path)
                assert cur_data.shape[0] >= n_points, "input pc_normal should have at least 4096 points"
                idx = np.random.choice(cur_data.shape[0], n_points, replace=False)
                cur_data = cur_data[idx]
                self.data.append({"pc_normal": cur_data, "uid": input_path.split("/")[-1].split(".")[0]})
        elif input_type == "mesh":
            # main.py:29-39 -> mesh_to_pc.py:42-57: load the file, draw n_points surface points + the normal of the face under each
            from .mesh_input import load_mesh, mesh_to_pc_normal
            if mc:
                raise NotImplementedError("--mc (mesh_to_pc.py:13-40: mesh2sdf signed distances + scikit-image marching cubes to make the "
                                          "input watertight before sampling) needs mesh2sdf and scikit-image, which are not installed")
            for input_path in input_list:
                vertices, faces = load_mesh(input_path)
                self.data.append({"pc_normal": mesh_to_pc_normal(vertices, faces, n_points), "uid": input_path.split("/")[-1].split(".")[0]})
        # any other value yields an empty dataset, like the reference's default 'pc' (main.py:70-75)
        print(f"dataset total data samples: {len(self.data)}")

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Dict:
        return {"pc_normal": normalize_pc(self.data[idx]["pc_normal"]), "uid": self.data[idx]["uid"]}
