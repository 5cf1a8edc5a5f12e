"""The library-level communicator (rsrec_comm_*: RCCL bound by librsrec itself, no MPI, no torch) and the device-side images it
reduces.  The GPU box has ONE device, and RCCL refuses two ranks on one device: what runs here is a one-rank communicator on the
LAST visible device (the whole RCCL path -- dlopen, unique id by value, communicator, all-reduce on the engine's stream -- with a
result that must be the identity), the id exchange through a file, and the image packers against host packing.  The N > 1 reduction
itself has no hardware record from this repository."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import objects_from, supercell_problem
from rslmtoasa_amd import _lib
from rslmtoasa_amd.recursion import Recursion

pytestmark = pytest.mark.gpu


def last_device():
    return _lib.lib().rsrec_device_count() - 1


def test_one_rank_communicator_on_the_last_device(tmp_path):
    p = supercell_problem((4, 4, 8))
    sites = np.array([1, 77, 128], dtype=np.int32)
    rec = Recursion(*objects_from(p, sites, 8, emin=-3.0, emax=1.8), device=last_device())
    assert rec.comm_size() == (0, 1)
    x = np.arange(1000, dtype=np.float64)
    rec.allreduce_sum(x)                                   # no communicator: the identity, like the reference without MPI
    assert np.array_equal(x, np.arange(1000.0))
    rec.comm_init(0, 1, Recursion.comm_unique_id())
    assert rec.comm_size() == (0, 1)
    rec.allreduce_sum(x)                                   # host array, staged through the device
    assert np.array_equal(x, np.arange(1000.0))
    # device image: a / b2 of 3 sites inside an image over 5 sites, reduced where it lies
    import torch
    rec.recur_b()
    img = torch.full((2, 5, 18, 8), 7.0, dtype=torch.float64, device="cuda:%d" % last_device())
    rec.pack_diag(1, 5, img[0].data_ptr(), img[1].data_ptr())
    rec.allreduce_sum(img.data_ptr(), img.numel())
    a_img = img[0].cpu().numpy()
    assert np.array_equal(a_img[1:4], rec.a[:8, :, :3, 0].transpose(2, 1, 0)) and not a_img[0].any() and not a_img[4].any()
    # a second communicator, id through a file
    rec.comm_init(0, 1, path=str(tmp_path / "rsrec_comm.id"))
    assert os.path.getsize(tmp_path / "rsrec_comm.id") == 128 and rec.comm_size() == (0, 1)
    rec.allreduce_sum(x)
    assert np.array_equal(x, np.arange(1000.0))
    rec.close()


def test_pack_moments_is_the_host_packing():
    p = supercell_problem((4, 4, 8))
    sites = np.array([5, 9, 33], dtype=np.int32)
    lld = 6
    rec = Recursion(*objects_from(p, sites, lld, emin=-3.0, emax=1.8), device=0)
    with pytest.raises(_lib.RsrecError):
        rec.pack_moments(0, 3, np.zeros((18, 18, 2 * lld + 2, 3), np.complex128, order="F"))      # nothing resident yet
    rec.chebyshev_recur()
    img = np.full((18, 18, 2 * lld + 2, 6), 3.0 + 1.0j, np.complex128, order="F")
    rec.pack_moments(2, 6, img)
    assert np.array_equal(img[:, :, :, 2:5], rec.mu_n[:, :, :, :3]) and not img[:, :, :, :2].any() and not img[:, :, :, 5:].any()
    rec.zsqr()                                             # scratch of other calls must not disturb the resident moments
    img2 = np.zeros_like(img)
    rec.pack_moments(2, 6, img2)
    assert np.array_equal(img2, img)
    with pytest.raises(_lib.RsrecError):
        rec.pack_moments(4, 6, img)                        # sites 5..7 of 6
    rec.close()
