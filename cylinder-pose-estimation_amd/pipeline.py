"""Whole hot path for a batch of stereo frames, resident on one GPU:
    left/right u8 [F,h,w]  ->  detect_grid (both images)  ->  fitSingleCylinder  ->  pose records [F,16] f64

This is the loop body of exp_gridDetection.m:55-81 (makePyGridPts L/R, then fitSingleCylinder per frame)
turned into batched kernel launches; frames are processed in chunks so that the detect workspace stays
bounded (about 170 MB per 1920x1200 image: DESIGN.md section 3)."""
import torch

from . import api, fit

REC = 16   # f64 per frame: cyl0[6] | cyl[6] (both after applyCylParamsPrior) | f0 f | meanError | packed counters


def pack_counters(n_pts, iters, fit_status, det_l, det_r):
    """n_pts (<4096), iterations (<2^20), statuses -> one exactly representable double"""
    code = n_pts.to(torch.int64) + 4096 * (iters.to(torch.int64) + (1 << 20) * (fit_status.to(torch.int64) * 64 +
                                                                                 det_l.to(torch.int64) * 8 + det_r.to(torch.int64)))
    return code.to(torch.float64)


def unpack_counters(code):
    code = code.to(torch.int64)
    n_pts = code % 4096
    rest = code // 4096
    iters = rest % (1 << 20)
    st = rest // (1 << 20)
    return n_pts, iters, st // 64, (st // 8) % 8, st % 8


class FramePipeline:
    """reusable workspaces for chunks of `chunk` stereo frames of size h x w"""

    def __init__(self, h, w, K1, K2, T21, radius, chunk=64, device='cuda:0', selector=fit.SEL_CHOOSE_IDX, th=0.3,
                 fit_mode=fit.FIT_NELDER_MEAD, lanes=1, ransac=None, stage='full'):
        self.h, self.w, self.chunk, self.device = h, w, chunk, torch.device(device)
        self.K1, self.K2, self.T21, self.radius = K1, K2, T21, radius
        self.selector, self.th, self.fit_mode = selector, th, fit_mode
        self.ransac = ransac            # None, or keywords of fit.fit_cylinder_ransac_batch (build-defined config 5)
        if stage not in ('full', 'detect'):
            raise ValueError("stage is 'full' or 'detect'")
        self.stage = stage              # 'detect': stop after chooseIdx + triangulate (fitSingleCylinder.m:12-17), no cylinder fit
        self.ws = {}
        # chunks are independent: `lanes` of them are in flight on their own HIP streams (each with its own
        # workspace), so the serial tails of one chunk (blob grouping, line fitting: one workgroup per frame)
        # run beside the wide kernels of the next
        self.lanes = max(1, int(lanes))
        self._streams = None

    def _ws(self, n_img, lane=0):
        """one workspace per lane, made for the first (largest) chunk it sees; smaller chunks are laid out inside it"""
        ws = self.ws.get(lane)
        if ws is None or not ws.fits(n_img, self.h, self.w):
            ws = self.ws[lane] = api.DetectWorkspace(n_img, self.h, self.w, self.device)
        return ws.use(n_img)

    @staticmethod
    def _interleaved(left, right):
        """left = S[:, 0], right = S[:, 1] of one contiguous stereo tensor S [F,2,h,w] (frame-major pairs)?"""
        h, w = left.shape[1:]
        return (left.dim() == 3 and left.shape == right.shape and left.stride() == (2 * h * w, w, 1) and right.stride() == left.stride()
                and right.data_ptr() == left.data_ptr() + h * w and left.untyped_storage().data_ptr() == right.untyped_storage().data_ptr())

    def run_chunk(self, left, right, lane=0, frame0=0):
        c = left.shape[0]
        if c > 0 and self._interleaved(left, right):
            # the pairs already lie one after the other in memory: the detect call reads them in place (L0 R0 L1 R1 ...)
            frames = torch.as_strided(left, (2 * c, self.h, self.w), (self.h * self.w, self.w, 1))
            det = api.detect_grid_batch(frames, self._ws(2 * c, lane))
            sl = lambda k, o: det[k][o::2].contiguous()
            g1 = fit.GridTables(sl('xy', 0), sl('id', 0), sl('n', 0))
            g2 = fit.GridTables(sl('xy', 1), sl('id', 1), sl('n', 1))
            st_l, st_r = det['status'][0::2], det['status'][1::2]
        else:
            frames = torch.cat([left, right])             # [2c,h,w]: one detect call for both cameras (a copy)
            det = api.detect_grid_batch(frames, self._ws(2 * c, lane))
            g1 = fit.GridTables(det['xy'][:c], det['id'][:c], det['n'][:c])
            g2 = fit.GridTables(det['xy'][c:], det['id'][c:], det['n'][c:])
            st_l, st_r = det['status'][:c], det['status'][c:]
        if self.stage == 'detect':      # BASELINE configs[1]: grid detection + triangulation only
            out = fit.select_triangulate_batch(g1, g2, self.K1, self.K2, self.T21, self.selector, 3, self.th)
            rec = torch.zeros((c, REC), dtype=torch.float64, device=frames.device)
            rec[:, 14] = out['mean_err']
            zero = torch.zeros_like(out['m'])
            rec[:, 15] = pack_counters(out['m'], zero, zero, st_l, st_r)
            return rec, det, out
        rk = None if self.ransac is None else dict(self.ransac, frame0=int(self.ransac.get('frame0', 0)) + frame0)
        out = fit.fit_single_cylinder_batch(g1, g2, self.K1, self.K2, self.T21, self.radius, self.selector, 3, self.th,
                                            ransac=rk, mode=self.fit_mode)
        rec = torch.empty((c, REC), dtype=torch.float64, device=frames.device)
        rec[:, 0:6] = out['cyl'][:, 0]
        rec[:, 6:12] = out['cyl'][:, 1]
        rec[:, 12:14] = out['fvals']
        rec[:, 14] = out['mean_err']
        rec[:, 15] = pack_counters(out['m'], out['iters'][:, 0], out['status'], st_l, st_r)
        return rec, det, out

    def run(self, left, right):
        """left/right: u8 [F,h,w] on the device -> records f64 [F,16]"""
        F = left.shape[0]
        recs = torch.empty((F, REC), dtype=torch.float64, device=left.device)
        n_chunks = (F + self.chunk - 1) // self.chunk
        if self.lanes == 1 or n_chunks == 1 or left.device.type != 'cuda':
            for i0 in range(0, F, self.chunk):
                i1 = min(F, i0 + self.chunk)
                recs[i0:i1] = self.run_chunk(left[i0:i1], right[i0:i1], 0, i0)[0]
            return recs
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=left.device) for _ in range(self.lanes)]
        main = torch.cuda.current_stream(left.device)
        for st in self._streams:
            st.wait_stream(main)                          # inputs and `recs` are ready on the caller's stream
        for k, i0 in enumerate(range(0, F, self.chunk)):
            i1 = min(F, i0 + self.chunk)
            lane = k % self.lanes
            with torch.cuda.stream(self._streams[lane]):
                recs[i0:i1] = self.run_chunk(left[i0:i1], right[i0:i1], lane, i0)[0]
        for st in self._streams:
            main.wait_stream(st)
        return recs
