"""PATH-like result of the Reeds-Shepp feasibility search (reference: src/env/reeds_shepp.py:11-19).

The GPU kernel returns only what the reference's consumers read -- `ctypes` and `lengths`
(src/model/agent/parking_agent.py:12-41).  `x / y / yaw / directions` are regenerated on demand on
the host by walking the word with the reference's sampling rule (0.1 m steps, segment end points,
generate_local_course :452-507) -- a convenience for plotting, not part of the hot path.
"""
import math

MAXC = math.tan(0.75) / 2.8           # car_parking_base.py:422
TYPE_NAMES = {0: 'S', 1: 'L', 2: 'R'}


def _wrap(t):
    while t > math.pi:
        t -= 2.0 * math.pi
    while t < -math.pi:
        t += 2.0 * math.pi
    return t


def _advance(l, kind, o):
    """pose reached after normalised arc/line parameter l from local origin o = (x, y, yaw)."""
    ox, oy, oyaw = o
    if kind == 'S':
        return ox + l / MAXC * math.cos(oyaw), oy + l / MAXC * math.sin(oyaw), oyaw
    ldx = math.sin(l) / MAXC
    ldy = (1.0 - math.cos(l)) / (MAXC if kind == 'L' else -MAXC)
    gdx = math.cos(-oyaw) * ldx + math.sin(-oyaw) * ldy
    gdy = -math.sin(-oyaw) * ldx + math.cos(-oyaw) * ldy
    return ox + gdx, oy + gdy, (oyaw + l if kind == 'L' else oyaw - l)


class PATH:
    def __init__(self, lengths, ctypes, start_pose, step=0.1):
        self.lengths = [float(v) for v in lengths]        # metres, signed (+ forward, - backward)
        self.ctypes = list(ctypes)                        # 'S' | 'L' | 'R'
        self.L = sum(abs(v) for v in self.lengths)
        self._start = tuple(float(v) for v in start_pose)
        self._step = step
        self._samples = None

    def _sample(self):
        if self._samples is not None:
            return self._samples
        norm = [v * MAXC for v in self.lengths]
        step = self._step * MAXC
        pts, dirs = [(0.0, 0.0, 0.0)], [1 if norm[0] > 0.0 else -1]
        origin = (0.0, 0.0, 0.0)
        ll = 0.0
        for i, (kind, l) in enumerate(zip(self.ctypes, norm)):
            d = step if l > 0.0 else -step
            pd = (-d - ll) if (i >= 1 and norm[i - 1] * l > 0) else (d - ll)
            first = True
            while abs(pd) <= abs(l):
                p = _advance(pd, kind, origin)
                if first and i >= 1:
                    pts[-1], dirs[-1] = p, (1 if pd > 0.0 else -1)   # overwrites the previous end point
                else:
                    pts.append(p)
                    dirs.append(1 if pd > 0.0 else -1)
                first = False
                pd += d
            ll = l - pd - d
            end = _advance(l, kind, origin)
            if first and i >= 1:
                pts[-1], dirs[-1] = end, (1 if l > 0.0 else -1)
            else:
                pts.append(end)
                dirs.append(1 if l > 0.0 else -1)
            origin = end
        sx, sy, syaw = self._start
        c, s = math.cos(-syaw), math.sin(-syaw)
        self._samples = ([c * x + s * y + sx for x, y, _ in pts], [-s * x + c * y + sy for x, y, _ in pts],
                         [_wrap(w + syaw) for _, _, w in pts], dirs)
        return self._samples

    x = property(lambda self: self._sample()[0])
    y = property(lambda self: self._sample()[1])
    yaw = property(lambda self: self._sample()[2])
    directions = property(lambda self: self._sample()[3])

    def __repr__(self):
        return 'PATH(' + ' '.join(f'{c}{l:+.3f}' for c, l in zip(self.ctypes, self.lengths)) + f', L={self.L:.3f})'
