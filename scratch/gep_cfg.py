import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import starneig_amd as S
S.node_init(1, 1, S.NO_MESSAGES)
torch.zeros(1, device='cuda')
n = int(sys.argv[1])
tH0, tR0 = S.device_matrix(n), S.device_matrix(n)
S.lcg_pencil_device(tH0, tR0, n)
for cfg in sys.argv[2:]:
    parts = [int(x) for x in cfg.split(",")]
    aed, ns, small = parts[:3]
    nib = parts[3] if len(parts) > 3 else -1
    tH, tR = tH0.clone(), tR0.clone()
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    conf = S.schur_init_conf(); conf.aed_window_size = aed; conf.shift_count = ns; conf.small_limit = small; conf.aed_nibble = nib
    torch.cuda.synchronize(); t = time.time()
    rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n, conf=conf)
    torch.cuda.synchronize(); dt = time.time() - t
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    print(cfg, 'rc', rc, '%.2fs' % dt, 'sweeps', st['sweeps'], 'aeds', st['aeds'], 'chase', st['chase_launches'],
          'gemmTF %.1f' % (st['gemm_flops'] / 1e12), 'aed_host %.2fs wait %.2fs' % (st['aed_host_s'], st['gpu_wait_s']),
          'res %.0f orthq %.0f orthz %.0f' % (ca['residual_u'], ca['orthogonality_q_u'], ca['orthogonality_z_u']), flush=True)
