import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import starneig_amd as S
S.node_init(1, 1, S.NO_MESSAGES)
torch.zeros(1, device='cuda')
n = int(sys.argv[1])
tH0, tR0 = S.device_matrix(n), S.device_matrix(n)
S.lcg_pencil_device(tH0, tR0, n)
prev = None
for rep in range(int(sys.argv[2])):
    tH, tR = tH0.clone(), tR0.clone()
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
    torch.cuda.synchronize()
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    _, cb = S.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    same = None if prev is None else (bool(torch.equal(prev[0], tH)), bool(torch.equal(prev[1], tR)))
    prev = (tH, tR)
    print(rep, 'rc', rc, 'sweeps', st['sweeps'], 'aeds', st['aeds'], 'resA %.0f resB %.0f' % (ca['residual_u'], cb['residual_u']),
          'orth %.0f %.0f' % (ca['orthogonality_q_u'], ca['orthogonality_z_u']), 'same as prev', same, flush=True)
