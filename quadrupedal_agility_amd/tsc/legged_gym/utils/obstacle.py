"""`Obstacle` -- the agility course generator of the task-level tree (tsc/legged_gym/utils/obstacle.py): per env six
obstacles (bar jump, A-frame, weave poles, see-saw, tyre jump, tunnel) placed along a fixed serpentine of six frames with
small random offsets, rasterised into ONE int16 height map (`height_field_raw`, 5 cm x 5 mm), an edge mask (`x_edge_mask`, for
the feet_edge reward) and 4 goals per obstacle (`env_goals`).  Same attributes and the same numbers as the reference class
(`tests/test_tsc_obstacle.py` holds it to arrays produced by the reference's own class); what is different is the structure:

  * an obstacle is DATA here -- a list of stamps (rectangles with a constant, ramp or arc profile), edge strips and goal
    offsets built by `_shape()` -- and one routine (`_place`) rotates / translates any of them into the env tile, instead of
    six drawing methods plus an inline transform;
  * the polygon rasteriser the reference takes from scikit-image (`skimage.draw.polygon`, not installed here) is restated
    as `fill_polygon`: every pixel centre inside or on the boundary of the polygon, inside the polygon's bounding box clipped
    to the tile;
  * randomness comes from a `random.Random` / `numpy.random.RandomState` pair owned by the object (seedable), consumed in the
    reference's order; with `seed=None` the module-level generators are used, exactly like the reference.

In this build the height map is not only what the height scan reads: it is also the collision terrain of the physics kernel
(terrain_type 1) -- the reference collides with separate obstacle meshes whose top surface the map
encodes.
"""
import random as _random

import numpy as np

TYPES = ("bar_jump", "frame", "poles", "seesaw", "tire_jump", "tunnel")
NO_CEILING = 32767          # include/qa_sim.h QA_NO_CEILING


def fill_polygon(rows, cols, shape):
    """Pixels (r, c) of a `shape` image whose centre lies inside or on the boundary of the polygon with vertices
    (rows[i], cols[i]) -- the contract of skimage.draw.polygon(r, c, shape): bounding box [floor(max(0, min)), ceil(max)]
    clipped to the image, boundary pixels included."""
    r = np.asarray(rows, dtype=np.float64); c = np.asarray(cols, dtype=np.float64)
    r0, r1 = int(max(0, r.min())), min(shape[0] - 1, int(np.ceil(r.max())))
    c0, c1 = int(max(0, c.min())), min(shape[1] - 1, int(np.ceil(c.max())))
    if r1 < r0 or c1 < c0:
        return np.zeros(0, dtype=np.intp), np.zeros(0, dtype=np.intp)
    rr, cc = np.meshgrid(np.arange(r0, r1 + 1), np.arange(c0, c1 + 1), indexing="ij")
    y, x = rr.astype(np.float64), cc.astype(np.float64)
    inside = np.zeros(rr.shape, dtype=bool)
    edge = np.zeros(rr.shape, dtype=bool)
    n = len(r)
    for i in range(n):
        ya, xa, yb, xb = r[i - 1], c[i - 1], r[i], c[i]
        # crossing number with the half-open rule (an edge owns its lower end point)
        straddle = (ya > y) != (yb > y)
        with np.errstate(divide="ignore", invalid="ignore"):
            xi = xa + (y - ya) * (xb - xa) / (yb - ya)
        inside ^= straddle & (x < xi)
        # on the closed segment: collinear and inside its bounding box
        cross = (xb - xa) * (y - ya) - (yb - ya) * (x - xa)
        on = (np.abs(cross) <= 1e-9 * max(1.0, abs(xb - xa) + abs(yb - ya)))
        on &= (x >= min(xa, xb) - 1e-12) & (x <= max(xa, xb) + 1e-12) & (y >= min(ya, yb) - 1e-12) & (y <= max(ya, yb) + 1e-12)
        edge |= on
    m = inside | edge
    return rr[m], cc[m]


class _Tile:
    """one env-sized raster: heights (int16), edge mask, the rectangles that carry content, goals"""

    def __init__(self, shape):
        self.h = np.zeros(shape, dtype=np.int16)
        self.ceil = np.full(shape, NO_CEILING, dtype=np.int16)      # undersides of overhangs (this build's addition, see Obstacle.ceiling_raw)
        self.edge = np.zeros(shape, dtype=bool)
        self.rects, self.edge_rects = [], []
        self.goals = None


def _corners(x0, y0, dx, dy):
    return [[x0, y0], [x0 + dx, y0], [x0 + dx, y0 + dy], [x0, y0 + dy]]


class Obstacle:
    def __init__(self, cfg, num_envs, seed=None, skip_envs=0):
        """`skip_envs`: this object builds envs [skip_envs, skip_envs + num_envs) of a larger job -- the generator's draws for the envs
        before them are consumed (not rasterised) first, so a rank of a data-parallel run gets exactly the obstacles the one-process job
        gives those envs, and the job is the same job at every world size (the sequential stream itself stays the reference's)"""
        self.cfg = cfg
        self.num_envs = self.num_robots = self.num_obstacles = num_envs
        self.num_cols = int(np.floor(np.sqrt(num_envs)))
        self.num_rows = int(np.ceil(num_envs / self.num_cols))
        self.env_length, self.env_width = cfg.env_length, cfg.env_width
        self.num_links_per_obst, self.num_joints_per_obst = cfg.num_obstacle_links, cfg.num_obstacle_joints
        self.proportions = [np.sum(cfg.obstacle_proportions[:i + 1]) for i in range(len(cfg.obstacle_proportions))]
        self.horizontal_scale, self.vertical_scale = cfg.horizontal_scale, cfg.vertical_scale
        self.num_goals = cfg.num_goals
        self.obst_types = list(cfg.obstacle_dict.keys())
        self.num_obst_per_env = cfg.num_obst_per_env
        self.frame_pos = np.array(cfg.frame_pos)
        self.frame_ang = np.radians(np.array(cfg.frame_ang))
        self.random_yaw = np.radians(cfg.random_yaw)
        self.seesaw_dof_pos = -np.arcsin(0.25 / 1.5)
        self.curriculum, self.curr_step, self.curr_threshold = cfg.curriculum, cfg.curr_step, cfg.curr_threshold
        self.bar_jump_joint_bias, self.tire_jump_joint_bias = -1, -10
        self.last_goal_repeat = cfg.last_goal_repeat
        if seed is None:
            self._py, self._np = _random, np.random
        else:
            self._py, self._np = _random.Random(seed), np.random.RandomState(seed)

        hs = self.horizontal_scale
        self.width_per_env_pixels, self.length_per_env_pixels = int(self.env_width / hs), int(self.env_length / hs)
        self.border = int(cfg.border_size / hs)
        self.tot_cols = int(self.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(self.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self.x_edge_mask = np.zeros((self.tot_rows, self.tot_cols), dtype=bool)
        # not in the reference: the undersides of the two overhanging obstacles (tunnel roof, upper arc of the tyre) on the same
        # grid, for this build's physics (QA_T_CEILING_SAMPLES) -- the reference collides with the obstacle meshes instead
        self.ceiling_raw = np.full((self.tot_rows, self.tot_cols), NO_CEILING, dtype=np.int16)

        xx, yy = np.meshgrid(np.arange(self.num_rows), np.arange(self.num_cols))
        self.spacing_x, self.spacing_y, self.env_boarder = cfg.env_length, cfg.env_width, cfg.env_boarder
        self.env_center = np.array([self.spacing_x, self.spacing_y]) / 2
        self.env_origins = np.zeros((num_envs, 3))
        self.env_origins[:, 0] = self.spacing_x * xx.flatten()[:num_envs]
        self.env_origins[:, 1] = self.spacing_y * yy.flatten()[:num_envs]
        k = self.num_obst_per_env
        self.obstacle_origins = np.zeros((num_envs, k, 3))
        self.obstacle_yaws = np.zeros((num_envs, k))
        self.obstacle_types = np.zeros((num_envs, k), dtype=int)
        self.obstacle_joint_pos = np.zeros((num_envs, k)) - 1          # -1: no joint
        self.env_goals = np.zeros((num_envs, k, cfg.num_goals, 3))
        for _ in range(int(skip_envs)):
            self._burn_env_draws()
        self._create_obstacle(list(range(num_envs)))

    # ------------------------------------------------------------------ shapes (obstacle.py:235-517), as data
    def _shape(self, kind, centre, joint_pos, true_joint=None):
        """-> _Tile with obstacle `kind` drawn axis-aligned around `centre` (metres, tile frame).  `true_joint`: the joint
        position without the below-ground marking offset (the ceiling field is not marked)"""
        hs, vs = self.horizontal_scale, self.vertical_scale
        t = _Tile((self.length_per_env_pixels, self.width_per_env_pixels))
        px, py = int(centre[0] / hs), int(centre[1] / hs)
        ratio = hs / vs
        g = np.zeros((self.num_goals, 3))
        zb = 0.3
        if kind == "bar_jump":
            w1, l1, h1 = int(1.2 / hs), int(0.2 / hs), int(joint_pos / vs)
            w2, l2, h2 = int(2.04 / hs), int(0.5 / hs), int(0.42 / vs)
            side = int((w2 - w1) / 2)
            for (x0, y0, dx, dy), v in (((int(px - l1 / 2), int(py - w1 / 2), l1, w1), h1),
                                        ((int(px - l2 / 2), int(py - w2 / 2), l2, side), h2),
                                        ((int(px - l2 / 2), int(py + w1 / 2), l2, side), h2)):
                t.h[x0:x0 + dx, y0:y0 + dy] = v
                t.rects.append(_corners(x0, y0, dx, dy))
            t.edge = t.h.astype(bool)
            X, Y = px * hs, py * hs
            g[:] = [[X - 1.8, Y, zb], [X - 0.9, Y, zb], [X, Y, joint_pos + zb], [X + 0.9, Y, zb]]
        elif kind in ("frame", "seesaw"):
            width = int(0.6 / hs)
            length = int((1.4625 if kind == "frame" else 1.5) / hs)
            height = int((0.333 if kind == "frame" else 0.26) / vs)
            slope = (height * vs) / (length * hs)
            ya, yb = int(py - width / 2), int(py + width / 2)
            ny = yb - ya + 1
            up = np.arange(int(px - length), int(px) + 1)
            dn = np.arange(int(px), int(px + length) + 1)
            t.h[up[0]:up[-1] + 1, ya:yb + 1] = np.tile((up - up[0]) * slope * ratio, (ny, 1)).T
            t.h[dn[0]:dn[-1] + 1, ya:yb + 1] = np.tile(np.flip(dn - dn[0]) * slope * ratio, (ny, 1)).T
            t.rects += [_corners(up[0], ya, length, width), _corners(dn[0], ya, length, width)]
            es = 1
            m = np.zeros_like(t.h)
            m[int(px - length):int(px + length) + 1, ya:ya + es + 1] = 1
            m[int(px - length):int(px + length) + 1, yb - es:yb + 1] = 1
            t.edge = m.astype(bool)
            t.edge_rects += [_corners(int(px - length), ya, 2 * length, es), _corners(int(px - length), yb - es, 2 * length, es)]
            X, Y, L, H = px * hs, py * hs, length * hs, height * vs
            g[:] = [[X - L - 0.7, Y, zb], [X - L, Y, zb], [X, Y, H + zb], [X + L, Y, zb]]
        elif kind == "poles":
            rad, height, dx = int((0.2 / 2) / hs), int(1.0 / vs), int(1.0 / hs)
            for i in range(4):
                x0, y0 = int(px - rad) + i * dx, int(py - rad)
                t.h[x0:x0 + 2 * rad, y0:y0 + 2 * rad] = height
                t.rects.append(_corners(x0, y0, 2 * rad, 2 * rad))
            X, Y, D = px * hs, py * hs, dx * hs
            for i in range(4):
                g[i] = [X + i * D, Y - 0.5 if i % 2 == 0 else Y + 0.5, zb]
        elif kind == "tire_jump":
            rad, width = int((0.8 / 2) / hs), int(1.5 / hs)
            l1, l2 = int(0.2 / hs), int(0.6 / hs)
            h1, h2 = int(joint_pos / vs), int(1.5 / vs)
            xs = np.arange(int(px - l1 / 2), int(px + l1 / 2) + 1)
            ys = np.arange(int(py - rad), int(py + rad) + 1)
            t.h[xs[0]:xs[-1] + 1, ys[0]:ys[-1] + 1] = np.tile(self.get_circle_height(ys - ys[0]) * ratio + h1, (len(xs), 1))
            # the tyre is a ring: the map above is its LOWER inner arc (hole bottom at joint - r), this is the upper one
            t.ceil[xs[0]:xs[-1] + 1, ys[0]:ys[-1] + 1] = np.tile(-self.get_circle_height(ys - ys[0]) * ratio + int((joint_pos if true_joint is None else true_joint) / vs), (len(xs), 1))
            t.h[int(px - l2 / 2):int(px + l2 / 2) + 1, int(py - width / 2):int(py - rad) + 1] = h2
            t.h[int(px - l2 / 2):int(px + l2 / 2) + 1, int(py + rad):int(py + width / 2) + 1] = h2
            t.rects += [_corners(int(px - l1 / 2), int(py - rad), l1, 2 * rad),
                        _corners(int(px - l2 / 2), int(py - width / 2), l2, width / 2 - rad),
                        _corners(int(px - l2 / 2), int(py + rad), l2, width / 2 - rad)]
            t.edge = t.h.astype(bool)
            X, Y = px * hs, py * hs
            g[:] = [[X - 1.8, Y, zb], [X - 0.9, Y, zb], [X, Y, joint_pos], [X + 0.9, Y, zb]]
        elif kind == "tunnel":
            rad, length = int((0.8 / 2) / hs), int(2.0 / hs)
            xs = np.arange(px, int(px + length) + 1)
            ys = np.arange(int(py - rad), int(py + rad) + 1)
            t.h[xs[0]:xs[-1] + 1, ys[0]:ys[-1] + 1] = np.tile((self.get_circle_height(ys - ys[0]) + rad) * ratio, (len(xs), 1))
            # the tunnel is a pipe of inner radius `rad`: the map above is its floor (lower half), this is the roof (upper half)
            t.ceil[xs[0]:xs[-1] + 1, ys[0]:ys[-1] + 1] = np.tile((-self.get_circle_height(ys - ys[0]) + rad) * ratio, (len(xs), 1))
            t.rects.append(_corners(px, int(py - rad), length, 2 * rad))
            X, Y = px * hs, py * hs
            g[:] = [[X - 1.0, Y, zb], [X - 0.5, Y, zb], [X + length * hs / 2, Y, zb], [X + length * hs + 0.5, Y, zb]]
        else:
            raise ValueError(kind)
        t.goals = g
        return t

    @staticmethod
    def get_circle_height(x_range):
        n = len(x_range) - 1
        return -np.sqrt((n / 2) ** 2 - (x_range - n / 2) ** 2)

    # ------------------------------------------------------------------ placement (obstacle.py:137-203)
    def _place(self, t, pivot, target, yaw):
        """rotate tile `t` by `yaw` about `pivot` and move the pivot to `target` (both in pixels): every content rectangle is
        mapped, filled as a polygon, and each covered pixel takes the value of the source pixel it came from"""
        rot = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        shape = t.h.shape
        out_h, out_e, out_c = np.zeros_like(t.h), np.zeros_like(t.edge), np.full_like(t.ceil, NO_CEILING)
        for rects, src, dst in ((t.rects, t.h, out_h), (t.edge_rects, t.edge, out_e), (t.rects, t.ceil, out_c)):
            # the reference writes the moved corners back into the corner array itself: where that array holds integers
            # (every shape but the tyre, whose side plates have a fractional width) the corners are truncated towards zero
            integral = len(rects) > 0 and np.issubdtype(np.array(rects).dtype, np.integer)
            for rect in rects:
                pts = np.array([rot @ (np.asarray(p, dtype=np.float64) - pivot) + target for p in rect])
                if integral:
                    pts = np.trunc(pts)
                rr, cc = fill_polygon(pts[:, 0], pts[:, 1], shape)
                r0 = np.clip(np.round((rr - target[0]) * np.cos(yaw) + (cc - target[1]) * np.sin(yaw) + pivot[0]).astype(int), 0, shape[0] - 1)
                c0 = np.clip(np.round((cc - target[1]) * np.cos(yaw) - (rr - target[0]) * np.sin(yaw) + pivot[1]).astype(int), 0, shape[1] - 1)
                dst[rr, cc] = src[r0, c0]
        return out_h, out_e, rot, out_c

    def add_border(self, mat):
        hs = self.horizontal_scale
        width, length = int(10.0 / hs), int(7.0 / hs)
        height, th = int(2.0 / self.vertical_scale), int(0.1 / hs)
        mat[0:length, 0:th] = height
        mat[0:th, 0:width] = height
        mat[length - th:length, 0:width] = height
        mat[0:length, width - th:width] = height

    def _burn_env_draws(self):
        """consume one env's draws in _create_obstacle's order: the shuffle, then per obstacle x / y / yaw noise and the bar / tyre height"""
        cfg = self.cfg
        order = list(self.obst_types)
        self._py.shuffle(order)
        for kind in order:
            rx = cfg.random_x[kind]
            self._np.uniform(rx[0], rx[1]); self._np.uniform(cfg.random_y[0], cfg.random_y[1])
            self._np.uniform(self.random_yaw[0], self.random_yaw[1])
            if kind in ("bar_jump", "tire_jump"):
                lo_hi = getattr(cfg, f"{kind}_init_range" if self.curriculum else f"{kind}_range")
                self._np.uniform(lo_hi[0], lo_hi[1])

    def _create_obstacle(self, env_ids):
        cfg, hs = self.cfg, self.horizontal_scale
        shape = (self.length_per_env_pixels, self.width_per_env_pixels)
        shift = {"poles": [-1.5, 0], "tunnel": [-1.0, 0]}
        for i in env_ids:
            order = list(self.obst_types)
            self._py.shuffle(order)
            tile_h = np.zeros(shape, dtype=np.int16)
            tile_c = np.full(shape, NO_CEILING, dtype=np.int16)
            tile_e = np.zeros(shape, dtype=np.int16)
            tile_goals = np.zeros((self.num_obst_per_env, self.num_goals, 3))
            for j, kind in enumerate(order):
                idx = self.obst_types.index(kind)
                rx = cfg.random_x[kind]
                noise = np.array([self._np.uniform(rx[0], rx[1]), self._np.uniform(cfg.random_y[0], cfg.random_y[1])])
                noise_yaw = self._np.uniform(self.random_yaw[0], self.random_yaw[1])
                pos = (self.frame_pos[j][1] - self.frame_pos[j][0]) / 2 + self.frame_pos[j][0] + noise
                yaw = self.frame_ang[j] + noise_yaw
                bias = np.array(shift.get(kind, [0, 0]), dtype=np.float64)
                centre = self.env_center.copy()
                joint, pos_z = -1, 0
                if kind in ("bar_jump", "tire_jump"):
                    lo_hi = getattr(cfg, f"{kind}_init_range" if self.curriculum else f"{kind}_range")
                    off = self.bar_jump_joint_bias if kind == "bar_jump" else self.tire_jump_joint_bias
                    joint = self._np.uniform(lo_hi[0], lo_hi[1]) + off      # offset marks the movable cells in the map (removed below)
                elif kind == "seesaw":
                    pos_z, joint = 0.26, self.seesaw_dof_pos
                pos = pos + bias
                centre = centre + bias
                t = self._shape(kind, centre, joint, true_joint=joint - (self.tire_jump_joint_bias if kind == "tire_jump" else 0))
                pivot, target = (centre - bias) / hs, (pos - bias) / hs
                new_h, new_e, rot, new_c = self._place(t, pivot, target, yaw)
                for m in range(self.num_goals):
                    t.goals[m, :2] = rot @ (t.goals[m, :2] - pivot * hs) + target * hs
                self.obstacle_types[i, j] = idx
                self.obstacle_origins[i, j] = self.env_origins[i] + np.append(pos, pos_z)
                self.obstacle_yaws[i, j] = yaw
                if kind == "bar_jump":
                    joint -= self.bar_jump_joint_bias
                elif kind == "tire_jump":
                    joint -= self.tire_jump_joint_bias
                self.obstacle_joint_pos[i, j] = joint
                self.add_border(new_h)
                tile_h |= new_h
                tile_c = np.minimum(tile_c, new_c)
                tile_e |= new_e.astype(np.int16)
                tile_goals[j] = t.goals
            self.add_terrain_to_map(tile_h, tile_e, tile_goals, i, tile_c)
        # the movable parts were drawn 1 m (bar) / 10 m (tyre) below ground so that they can be told apart in the map
        vs = self.vertical_scale
        self.bar_jump_mask = (self.height_field_raw > int(-5 / vs)) & (self.height_field_raw < 0)
        self.tire_jump_mask = self.height_field_raw < int(-5 / vs)
        self.bar_jump_goal_mask = (self.env_goals > int(-5 / vs)) & (self.env_goals < 0)
        self.tire_jump_goal_mask = self.env_goals < int(-5 / vs)
        self.height_field_raw[self.bar_jump_mask] -= int(self.bar_jump_joint_bias / vs)
        self.height_field_raw[self.tire_jump_mask] -= int(self.tire_jump_joint_bias / vs)
        self.env_goals[self.bar_jump_goal_mask] -= self.bar_jump_joint_bias
        self.env_goals[self.tire_jump_goal_mask] -= self.tire_jump_joint_bias

    def add_terrain_to_map(self, tile_h, tile_e, tile_goals, i, tile_c=None):
        hs = self.horizontal_scale
        sx = int(self.border + self.env_origins[i, 0] / hs)
        sy = int(self.border + self.env_origins[i, 1] / hs)
        ex = int(self.border + self.env_origins[i, 0] / hs + self.length_per_env_pixels)
        ey = int(self.border + self.env_origins[i, 1] / hs + self.width_per_env_pixels)
        self.height_field_raw[sx:ex, sy:ey] = tile_h
        self.x_edge_mask[sx:ex, sy:ey] = tile_e
        if tile_c is not None:
            self.ceiling_raw[sx:ex, sy:ey] = tile_c
        self.env_goals[i] = tile_goals + np.array([self.env_origins[i, 0], self.env_origins[i, 1], 0])

    # ------------------------------------------------------------------ what the env turns the arrays into
    def flat_goals(self):
        """(num_envs, num_obst * num_goals + last_goal_repeat, 3): goals in course order, the last one repeated with
        y + 0.1 (k + 1) (tsc/legged_gym/envs/base/legged_robot.py:946-957)"""
        g = self.env_goals.reshape(self.num_envs, -1, 3)
        tail = []
        for k in range(self.last_goal_repeat):
            last = g[:, -1:, :].copy()
            last[:, :, 1] += 0.1 * (k + 1)
            tail.append(last)
        return np.concatenate([g] + tail, axis=1)
