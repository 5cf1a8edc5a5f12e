"""The small-batch level loop as a HIP graph (option `graph`): every SCF call of the reference runs recur_b on <= 4 sites, 19-49
levels x 6-8 dependent launches each (crecal_b, recursion.f90:1873-1973).  The captured graph must be invisible in the results:
bitwise equal to the launch-per-kernel path on the first call (capture + launch), on replays, and after the operator's VALUES change
(the nodes hold pointers, not blocks); a changed lattice, depth or seed list must re-capture."""
import numpy as np
import pytest

from helpers import RTOL, load_golden_with_inputs, objects_from, problem_dict, rel_err, supercell_problem
from rslmtoasa_amd.recursion import Recursion

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hoh", [False, True])
@pytest.mark.parametrize("spmm5", [1, 2])
def test_graph_replay_is_bitwise_the_launch_path(hoh, spmm5):
    p = supercell_problem((6, 6, 6), hoh=hoh)
    sites = np.array([1, 100, 216], dtype=np.int32)
    rec = Recursion(*objects_from(p, sites, 14), device=0)
    rec.set_option("spmm5", spmm5)
    rec.set_option("graph", 0)
    rec.recur_b()
    a0, b0 = rec.a_b.copy(), rec.b2_b.copy()
    rec.set_option("graph", 1)
    for _ in range(3):                      # capture + launch, then two replays
        rec.a_b[:] = 0
        rec.b2_b[:] = 0
        rec.recur_b()
        assert np.array_equal(rec.a_b, a0) and np.array_equal(rec.b2_b, b0)
    # new operator values, same pointers: the replayed graph must see them
    rec.hamiltonian.ee = rec.hamiltonian.ee * 1.25
    rec.update_hamiltonian()
    rec.recur_b()
    a1 = rec.a_b.copy()
    assert np.abs(a1 - a0).max() > 1e-3
    rec.set_option("graph", 0)
    rec.recur_b()
    assert np.array_equal(rec.a_b, a1)
    rec.close()


def test_graph_recaptures_when_depth_or_seeds_change(oracle_lib):
    p = supercell_problem((4, 4, 8))
    o = oracle_lib.Oracle(p)
    rec = Recursion(*objects_from(p, np.array([1, 77], dtype=np.int32), 10), device=0)
    rec.recur_b()
    rec.recur_b()
    a_o, b_o = o.block_lanczos(np.array([1, 77], dtype=np.int32), 10)
    assert rel_err(rec.a_b, a_o) < RTOL and rel_err(rec.b2_b, b_o) < RTOL
    rec.close()
    rec = Recursion(*objects_from(p, np.array([5, 9, 33], dtype=np.int32), 7), device=0)
    rec.recur_b()
    rec.recur_b()
    a_o, b_o = o.block_lanczos(np.array([5, 9, 33], dtype=np.int32), 7)
    assert rel_err(rec.a_b, a_o) < RTOL and rel_err(rec.b2_b, b_o) < RTOL
    rec.close()


def test_graph_recaptures_when_the_lattice_changes_on_the_same_handle():
    """Same handle, same kk / seeds / depth, another neighbour table (a vacancy: every reference to one atom removed) and then another
    nmax: the captured graph holds the region lists, grids and lattice dimensions by value, and a freed region entry or table commonly
    comes back at the same address -- the graph must go with the lattice, not be recognised by pointers (round-3 advisor finding)."""
    p = supercell_problem((6, 6, 6))
    sites = np.array([1, 100, 216], dtype=np.int32)
    rec = Recursion(*objects_from(p, sites, 12), device=0)

    def both_paths():
        out = []
        for graph in (1, 1, 0):
            rec.set_option("graph", graph)
            rec.a_b[:] = 0
            rec.recur_b()
            out.append((rec.a_b.copy(), rec.b2_b.copy()))
        assert np.array_equal(out[0][0], out[2][0]) and np.array_equal(out[0][1], out[2][1])
        assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1], out[2][1])
        return out[2]

    a0, _ = both_paths()
    nn = p["nn"].copy()
    nn[:, 1:][nn[:, 1:] == 50] = 0                       # atom 50 becomes unreachable: same kk, same columns, other regions
    rec.lattice.nn = nn
    rec.update_lattice()
    rec.update_hamiltonian()
    a1, _ = both_paths()
    assert np.abs(a1 - a0).max() > 1e-6
    # per-atom blocks for the first 3 atoms (copies of the stencil): other operator classes, same tables otherwise
    rec.lattice.nmax = 3
    rec.hamiltonian.hall = np.repeat(np.asarray(p["ee"])[:, :, :, :1], 3, axis=3) * 1.1
    rec.update_lattice()
    rec.update_hamiltonian()
    a2, _ = both_paths()
    assert np.abs(a2 - a1).max() > 1e-6
    for key, val in (("s5_waves", 4), ("batch", 2)):     # options whose value a captured launch holds
        rec.set_option(key, val)
        both_paths()
    rec.close()


def test_allreduce_sum_counts_complex_images_in_doubles():
    """One-rank communicator = identity; what is checked is the marshalling: a complex128 F-ordered image is 2 doubles per element, and
    anything that is not float64 / complex128 is refused (round-3 advisor finding: operator precedence let every F-ordered array through)."""
    p = supercell_problem((4, 4, 8))
    rec = Recursion(*objects_from(p, np.array([1], dtype=np.int32), 5), device=0)
    try:
        rec.comm_init(0, 1, Recursion.comm_unique_id())
    except Exception as e:                                # RCCL not loadable on this box: nothing to marshal into
        rec.close()
        pytest.fail("library communicator could not be created: %s" % e)
    img = np.asfortranarray((np.arange(18 * 18 * 4).reshape(18, 18, 4) * (1 + 2j)).astype(np.complex128))
    ref = img.copy()
    rec.allreduce_sum(img)
    assert np.array_equal(img, ref)
    with pytest.raises(TypeError):
        rec.allreduce_sum(np.zeros((4, 4), np.float32, order="F"))
    with pytest.raises(ValueError):
        rec.allreduce_sum(np.zeros((8, 8))[::2])
    rec.close()


@pytest.mark.parametrize("name", ["B2FeCo_block_hoh", "fccCu001_block_hoh"])
def test_graph_on_the_reference_cases(name):
    """The reference's own impurity / surface SCF cases (2 sites, per-atom hall blocks resp. 3 types): graph path vs golden coefficients."""
    g = load_golden_with_inputs(name)
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], int(g["lld"]), nsp=int(g["nsp"])), device=0)
    for _ in range(2):
        rec.recur_b()
        assert rel_err(rec.a_b, g["a_b"]) < RTOL and rel_err(rec.b2_b, g["b2_b"]) < RTOL
    rec.close()


@pytest.mark.parametrize("name", ["B2FeCo_block_hoh", "B2FeCo_block", "fccCu001_block_hoh", "bccFe_nsp4_block", "bccFe_nsp2_block_hoh"])
def test_device_assembled_operator_streams_equal_the_host_swizzle(name):
    """rsrec_set_hamiltonian assembles k_spmm5's operator streams on the device from the raw blocks (k_s5_emit; SURVEY 8 f2, our side of
    the boundary: 6.8 ms of host swizzle per call on the 18 operator classes of B2FeCo, more than its recursion call).  Option
    s5_host_emit = 1 keeps the host swizzle: both must give bit-identical coefficients (the streams are the same numbers), also after
    the operator's values change while its block structure stays (the cached-schedule path of every later SCF iteration)."""
    g = load_golden_with_inputs(name)
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], int(g["lld"]), nsp=int(g["nsp"])), device=0)
    rec.set_option("spmm5", 2)
    rec.set_option("graph", 0)
    out = {}
    for scale in (1.0, 0.9):
        rec.hamiltonian.ee = g["ee"] * scale
        for host in (1, 0, 0):
            rec.set_option("s5_host_emit", host)
            rec.update_hamiltonian()
            rec.recur_b()
            out.setdefault(scale, []).append((rec.a_b.copy(), rec.b2_b.copy()))
        (a1, b1), (a2, b2), (a3, b3) = out[scale]
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(a1, a3) and np.array_equal(b1, b3)
    assert rel_err(out[1.0][1][0], g["a_b"]) < RTOL and rel_err(out[1.0][1][1], g["b2_b"]) < RTOL
    assert np.abs(out[0.9][1][0] - out[1.0][1][0]).max() > 1e-4
    rec.close()
