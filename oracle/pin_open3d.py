#!/usr/bin/env python3
"""Pin-on-first-contact script for the trimesh / open3d rules behind `object_width` (TEST INFRASTRUCTURE ONLY).

predict.py:131-142 derives the crop-window size from the model when dataset_info has no 'object_width':
    mesh = trimesh.load(model_path); cloud = toOpen3dCloud(mesh.vertices).voxel_down_sample(0.005)
    object_width = (1 + boundingbox / 100) * compute_cloud_diameter(cloud.points) * 1000
so two third-party rules feed EVERY integer bbox of a real run: which vertices trimesh hands over (trimesh.load defaults to
process=True: duplicate / unreferenced vertices of a mesh with faces are merged, a point cloud is left alone) and how open3d's
voxel grid is anchored and averaged.  Neither package exists in the offline build container: iros20-6d-pose-tracking_amd/utils.py
restates both (vertices as stored in the file; grid origin = min_bound - voxel / 2, mean per voxel) -- "parity unpinned", listed in
DESIGN.md section 4.  Run THIS script once where trimesh and open3d are installed (the reference's docker image):

    python oracle/pin_open3d.py [model.ply ...]     # writes tests/golden/open3d_object_width.npz
    python -m pytest tests/test_pinned_third_party.py

Stored per model: vertex count after trimesh.load with process=True / False, the down-sampled cloud (sorted), the diameter.
Default models: the fixtures' icosphere written as a .ply with faces and duplicated vertices, and (when present) the reference's
own object_models/bunny/1.ply."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as Fx  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "open3d_object_width.npz")
BUNNY = "/root/reference/object_models/bunny/1.ply"


def duplicated_mesh_ply(path, subdiv=2, radius=0.05, seed=3):
    """an icosphere whose first 20 vertices are stored twice (faces keep pointing at the first copy): trimesh's process=True
    drops the unreferenced copies, process=False keeps them"""
    m = Fx.icosphere(subdiv, radius, seed)
    v = np.concatenate([m["vertices"], m["vertices"][:20]])
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(m["faces"])))
        for p in v:
            f.write("%.9g %.9g %.9g\n" % tuple(p))
        for t in m["faces"]:
            f.write("3 %d %d %d\n" % tuple(t))
    return path


def pin(models):
    import open3d as o3d
    import trimesh
    from scipy.spatial import ConvexHull, distance_matrix
    out = {"trimesh_version": np.array(trimesh.__version__), "open3d_version": np.array(o3d.__version__), "models": np.array([m[0] for m in models])}
    for name, path in models:
        for proc in (True, False):
            mesh = trimesh.load(path, process=proc)
            v = np.asarray(mesh.vertices)
            cloud = o3d.geometry.PointCloud()
            cloud.points = o3d.utility.Vector3dVector(v.astype(np.float64))
            ds = np.asarray(cloud.voxel_down_sample(voxel_size=0.005).points)
            ds = ds[np.lexsort((ds[:, 2], ds[:, 1], ds[:, 0]))]
            hull = ConvexHull(ds)
            hp = ds[hull.vertices]
            tag = "%s_process%d" % (name, int(proc))
            out[tag + "_n_vertices"] = np.array(len(v))
            out[tag + "_cloud"] = ds
            out[tag + "_diameter_mm"] = np.array(float(np.max(distance_matrix(hp, hp))) * 1000)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        models = [("icosphere_dup", duplicated_mesh_ply(os.path.join(tmp, "dup.ply")))]
        if os.path.isfile(BUNNY):
            models.append(("bunny", BUNNY))
        models += [(os.path.splitext(os.path.basename(p))[0], p) for p in sys.argv[1:]]
        pin(models)
