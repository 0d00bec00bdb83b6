"""CPU: radar .bin frames and tracking-result files (ratrack_amd/vod_io.py) -- formats of vod/frame/data_loader.py:164-180
and main_utils.py:165-184.  The result-line check restates the reference's string building literally."""
import numpy as np
import pytest
import torch

from ratrack_amd import vod_io


def test_radar_bin_round_trip_and_frame_tensors(tmp_path):
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(242, 7)).astype(np.float32), rng.normal(size=(322, 7)).astype(np.float32)
    pa, pb = tmp_path / "00001.bin", tmp_path / "00000.bin"
    vod_io.save_radar_bin(pa, a)
    vod_io.save_radar_bin(pb, b)
    sa, sb = vod_io.load_radar_bin(pa), vod_io.load_radar_bin(pb)
    assert sa.dtype == np.float32 and np.array_equal(sa, a) and np.array_equal(sb, b)
    pc1, pc2, f1, f2 = vod_io.frame_pair_tensors(sa, sb)
    assert pc1.shape == (1, 3, 242) and pc2.shape == (1, 3, 322) and f1.shape == (1, 2, 242) and f2.shape == (1, 2, 322)
    # main_utils.py:75-78: permute(0,2,1)[:, :3] of columns 0:3 and [:, 0:2] of columns 3:6
    assert np.array_equal(pc1[0].numpy(), a[:, :3].T) and np.array_equal(f2[0].numpy(), b[:, 3:5].T)
    with open(tmp_path / "bad.bin", "wb") as f:
        f.write(b"\0" * 40)
    with pytest.raises(ValueError):
        vod_io.load_radar_bin(tmp_path / "bad.bin")
    with pytest.raises(FileNotFoundError):
        vod_io.load_radar_bin(tmp_path / "missing.bin")


def test_ego_motion_compensation_matches_reference_expression():
    rng = np.random.default_rng(1)
    xyz = rng.normal(size=(50, 3))
    c, s = np.cos(0.03), np.sin(0.03)
    ego = np.array([[c, -s, 0, -0.8], [s, c, 0, 0.02], [0, 0, 1, 0], [0, 0, 0, 1]])
    out = vod_io.compensate_ego_motion(xyz, ego)
    hom = np.hstack((xyz, np.ones((50, 1))))
    assert np.allclose(out, (np.linalg.inv(ego) @ hom.T).T)            # [x 1] inv(E^T) == (inv(E) [x 1]^T)^T
    assert np.allclose(out[:, 3], 1.0)


def test_track_result_lines_match_reference_format(tmp_path):
    torch.manual_seed(0)
    objects = {7: torch.randn(1, 139, 3), 12: torch.randn(1, 139, 1)}
    confs = torch.tensor([0.75, 0.3333333])
    path = vod_io.write_track_results(str(tmp_path), "delft_1", 42, objects, confs)
    assert path.endswith("delft_1/00042.txt")
    lines = open(path).read().splitlines()
    # the reference's own construction (main_utils.py:170-182)
    want = []
    for idx, (obj_id, obj) in enumerate(objects.items()):
        s = "NA" + " 1" + " -1" + " -1" + " " + str(float(confs[idx])) + " " + str(obj_id)
        for i in range(obj.size(2)):
            s += " " + str(float(obj[0, 3, i])) + " " + str(float(obj[0, 4, i])) + " " + str(float(obj[0, 5, i]))
        want.append(s)
    assert lines == want
    back = vod_io.read_track_results(path)
    assert [b[0] for b in back] == [7, 12] and back[0][2].shape == (3, 3) and back[1][2].shape == (1, 3)
    assert np.allclose(back[0][2], objects[7][0, 3:6].T.double().numpy())
    assert abs(back[1][1] - float(confs[1])) < 1e-12


def test_read_reference_result_files_round_trip():
    """Three of the tracking-result files the reference ships (src/result/4dmot_runthis/delft_1/, copied as data into
    tests/golden/result_files/): parsed by read_track_results and written back by write_track_results BYTE FOR BYTE."""
    import os
    import tempfile
    import torch
    from _util import GOLDEN
    from ratrack_amd import vod_io
    d = os.path.join(GOLDEN, "result_files")
    for name in sorted(os.listdir(d)):
        src = os.path.join(d, name)
        rows = vod_io.read_track_results(src)
        assert rows and all(pts.shape[1] == 3 and pts.shape[0] >= 1 for _, _, pts in rows)
        objects, confs = {}, []
        for obj_id, conf, pts in rows:
            t = torch.zeros(1, 6, pts.shape[0], dtype=torch.float32)
            t[0, 3:6] = torch.from_numpy(pts.T).float()
            objects[obj_id] = t
            confs.append(conf)
        with tempfile.TemporaryDirectory() as tmp:
            out = vod_io.write_track_results(tmp, "delft_1", int(name.split("_")[-1][:5]), objects, confs)
            assert open(out).read() == open(src).read(), name
