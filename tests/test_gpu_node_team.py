"""GPU: several GPUs driven from ONE process through the plain StarNEig C interface --
starneig_node_init(cores, gpus = N, flags) followed by starneig_SEP_SM_Hessenberg / _Schur on host
arrays (reference common/node.c:200-216, :435-543: its shared-memory mode hands every CUDA device
of the node to the calling process).  csrc/node_team.hip runs one host thread per device.  On the
one-GPU test box the N ranks are virtual (STARNEIG_AMD_VIRTUAL_GPUS: N threads sharing cuda:0, the
collectives through the in-process exchange); on N real devices the same code takes RCCL."""
import os

import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, elementwise_tolerance

pytestmark = pytest.mark.gpu


@pytest.fixture
def team(node):
    """re-initialises the node with `gpus` virtual ranks; the single-GPU node comes back afterwards"""
    def start(gpus, cores=4):
        node.node_finalize()
        os.environ["STARNEIG_AMD_VIRTUAL_GPUS"] = str(gpus)
        node.node_init(cores, gpus, node.NO_MESSAGES)
        assert node.lib.load().starneig_node_get_gpus() == gpus
        return node
    yield start
    node.node_finalize()
    os.environ.pop("STARNEIG_AMD_VIRTUAL_GPUS", None)
    node.node_init(1, 1, node.NO_MESSAGES)
    assert node.lib.load().starneig_node_get_gpus() == 1


@pytest.mark.parametrize("gpus,n", [(2, 700), (3, 1100), (4, 2000)])
def test_c_interface_on_several_gpus(team, gpus, n):
    S = team(gpus)
    A0 = O.random_fullpos(n)
    A = A0.copy(order="F"); Q = O.identity(n)
    assert S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0]) == 0
    # against the oracle, elementwise, like the single-GPU path (tests/test_gpu_hessenberg.py)
    Ao = A0.copy(order="F"); Qo = O.identity(n)
    O.hessenberg(Ao, Qo)
    assert O.count_below_subdiagonal(A) == 0
    assert np.array_equal(np.sign(np.diag(A[:n], -1)), np.sign(np.diag(Ao[:n], -1)))
    assert np.abs(A[:n] - Ao[:n]).max() / np.linalg.norm(A0[:n]) <= elementwise_tolerance(n)
    assert O.residual_u(Q, A, A0) < 1.5 * 15 and O.orthogonality_u(Q) < 1.5 * 11
    H0 = A.copy(order="F")
    real = np.zeros(n); imag = np.zeros(n)
    assert S.SEP_SM_Schur(n, A, A.shape[0], Q, Q.shape[0], real, imag) == 0
    assert O.check_schur_form(A) == 0
    assert O.residual_u(Q, A, A0) < WARN_U and O.orthogonality_u(Q) < WARN_U
    er, ei = O.extract_eigenvalues(A)
    hook = O.eigenvalues_check((er, ei, np.ones(n)), (real, imag, np.ones(n)))
    assert hook["failures"] == 0 and hook["warnings"] == 0, hook
    Ho = H0.copy(order="F"); Zo = O.identity(n, ld=H0.shape[0])
    wro, wio = O.schur(Ho, Zo)
    assert O.match_eigenvalues(real + 1j * imag, wro + 1j * wio) < 1e4


@pytest.mark.parametrize("gpus,n", [(2, 1500), (3, 3000)])
def test_team_schur_equals_the_single_gpu_reduction_bit_for_bit(team, node, gpus, n):
    """every rank reduces a replica of H but updates only its rows of Q and its own column tiles of the
    deflated part of H (schur_update_pair_sharded_kernel): what comes back, assembled from the owners,
    is the single-GPU result bit for bit (n = 3000 has look-ahead sweeps: phase A and phase B)"""
    H0 = O.random_fullpos(n)
    Q0 = O.identity(n)
    assert node.SEP_SM_Hessenberg(n, H0, H0.shape[0], Q0, Q0.shape[0]) == 0
    H1 = H0.copy(order="F"); Q1 = Q0.copy(order="F"); r1 = np.zeros(n); i1 = np.zeros(n)
    assert node.SEP_SM_Schur(n, H1, H1.shape[0], Q1, Q1.shape[0], r1, i1) == 0
    S = team(gpus)
    H2 = H0.copy(order="F"); Q2 = Q0.copy(order="F"); r2 = np.zeros(n); i2 = np.zeros(n)
    assert S.SEP_SM_Schur(n, H2, H2.shape[0], Q2, Q2.shape[0], r2, i2) == 0
    assert np.array_equal(r1, r2) and np.array_equal(i1, i2)
    assert np.array_equal(Q1[:n], Q2[:n])
    bad = np.argwhere(H1[:n] != H2[:n])
    assert bad.size == 0, (len(bad), bad[:5], "columns / 128:", np.unique(bad[:, 1] // 128)[:10])


def test_gpus_request_is_clamped_and_partial_ranges_stay_on_one_device(team):
    S = team(2)
    L = S.lib.load()
    L.starneig_node_set_gpus(7)                 # more than there are: min(requested, present), node.c:200-216
    assert L.starneig_node_get_gpus() == 2
    L.starneig_node_set_gpus(1)
    assert L.starneig_node_get_gpus() == 1
    L.starneig_node_set_gpus(-1)                # STARNEIG_USE_ALL
    assert L.starneig_node_get_gpus() == 2
    n = 600
    A0 = O.random_fullpos(n)
    A = A0.copy(order="F"); Q = O.identity(n)
    conf = S.hessenberg_init_conf()
    assert S.SEP_SM_Hessenberg_expert(conf, n, n // 4, 3 * n // 4, A, A.shape[0], Q, Q.shape[0]) == 0
    Ao = A0.copy(order="F"); Qo = O.identity(n)
    O.hessenberg(Ao, Qo, begin=n // 4, end=3 * n // 4)
    assert np.abs(A[:n] - Ao[:n]).max() / np.linalg.norm(A0[:n]) <= elementwise_tolerance(n)
