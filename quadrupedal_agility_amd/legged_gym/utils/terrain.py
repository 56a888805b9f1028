"""Terrain tiles for the BBC task (API of bbc/legged_gym/utils/terrain.py:9-139).

The reference builds a 10 x 10 grid of 10 m tiles with isaacgym.terrain_utils generators and hands the
int16 height field (or its triangulation) to PhysX.  Here the same int16 field goes into the engine's
HEIGHT_SAMPLES tensor (include/qa_sim.h); 'heightfield' and 'trimesh' are the same surface for this
build (two triangles per cell; the trimesh slope-threshold correction does not trigger below 0.75,
and the BBC tiles stay under 0.16).  Only the tile types with a non-zero proportion in the Go2 config
are implemented: smooth pyramid slope and rough (uniform-noise) pyramid slope.  Random draws use
numpy's global RNG like the reference (seeded by set_seed), so tile layouts are distribution-, not
bit-compatible (isaacgym.terrain_utils is not available to compare against)."""
import numpy as np
from scipy.interpolate import RegularGridInterpolator


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale, self.horizontal_scale = vertical_scale, horizontal_scale
        self.width, self.length = width, length
        self.height_field_raw = np.zeros((width, length), dtype=np.int16)


def pyramid_sloped_terrain(terrain, slope=1.0, platform_size=1.0):
    """Pyramid rising (slope > 0) or sinking (slope < 0) towards the tile centre, cut off by a flat platform."""
    cx, cy = terrain.width // 2, terrain.length // 2
    fx = ((cx - np.abs(cx - np.arange(terrain.width))) / cx).reshape(terrain.width, 1)
    fy = ((cy - np.abs(cy - np.arange(terrain.length))) / cy).reshape(1, terrain.length)
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (peak * fx * fy).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    edge = terrain.height_field_raw[cx - half, cy - half]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def random_uniform_terrain(terrain, min_height, max_height, step=1.0, downsampled_scale=None):
    """Heights drawn from {min, min+step, ..., max} on a coarse grid, bilinearly upsampled and ADDED to the tile."""
    ds = terrain.horizontal_scale if downsampled_scale is None else downsampled_scale
    lo, hi, st = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale), int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    nx = int(terrain.width * terrain.horizontal_scale / ds)
    ny = int(terrain.length * terrain.horizontal_scale / ds)
    coarse = np.random.choice(levels, (nx, ny))
    x = np.linspace(0, terrain.width * terrain.horizontal_scale, nx)
    y = np.linspace(0, terrain.length * terrain.horizontal_scale, ny)
    f = RegularGridInterpolator((x, y), coarse.astype(np.float64), method="linear")
    xu = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width)
    yu = np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    gx, gy = np.meshgrid(xu, yu, indexing="ij")
    terrain.height_field_raw += np.rint(f(np.stack([gx, gy], axis=-1))).astype(np.int16)
    return terrain


class Terrain:
    def __init__(self, cfg, num_robots):
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ["none", "plane", None]:
            return
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self.difficulties = cfg.difficulties
        if cfg.curriculum:
            for j in range(cfg.num_cols):
                for i in range(cfg.num_rows):
                    self.add_terrain_to_map(self.make_terrain(j / cfg.num_cols + 0.001, i / cfg.num_rows), i, j)
        elif cfg.selected:
            raise NotImplementedError("terrain.selected is not used by the Go2 task")
        else:
            for k in range(cfg.num_sub_terrains):
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                choice = np.random.uniform(0, 1)
                difficulty = np.random.choice(self.difficulties)
                self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw

    def make_terrain(self, choice, difficulty):
        t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                       vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)
        slope = difficulty * 0.4
        if choice < self.proportions[0]:
            if choice < self.proportions[0] / 2:
                slope *= -1
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
        elif choice < self.proportions[1]:
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
            random_uniform_terrain(t, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        else:
            raise NotImplementedError("stairs / discrete / stepping-stone / gap / pit tiles have zero proportion in the Go2 config "
                                      "(terrain_proportions = [0.2, 0.8, 0, 0, 0]) and are not implemented")
        return t

    def add_terrain_to_map(self, terrain, row, col):
        sx, sy = self.border + row * self.length_per_env_pixels, self.border + col * self.width_per_env_pixels
        self.height_field_raw[sx:sx + self.length_per_env_pixels, sy:sy + self.width_per_env_pixels] = terrain.height_field_raw
        hs = terrain.horizontal_scale
        x1, x2 = int((self.env_length / 2.0 - 1) / hs), int((self.env_length / 2.0 + 1) / hs)
        y1, y2 = int((self.env_width / 2.0 - 1) / hs), int((self.env_width / 2.0 + 1) / hs)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]
