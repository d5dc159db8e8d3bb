"""A second, independent reading of the reference's BVH builder (src/bvh/bvh.cpp:31-249, src/bvh/bvh.hpp,
src/intersectables/aabb.hpp, src/constants.hpp:31-32) in numpy float32 — test infrastructure only.
Written from the reference's source text, not from nova_amd/csrc/host/scene_build.cpp: it checks the CPU
builder that in turn checks the device build.  Float32 arithmetic with a fixed operation order is
reproducible in numpy, so the two must agree bit for bit (the sign of a zero aside)."""
import numpy as np

F = np.float32
FLT_MAX = F(np.finfo(np.float32).max)
MIN_TRIANGLES_PER_LEAF = 8   # constants.hpp:31
MAX_BINS = F(1024.0)         # constants.hpp:32


def surface_area(top, bottom):  # aabb.hpp: dot(xyz(dims), yzx(dims)) * 2, glm::dot sums x, y, z in order
    d = top - bottom
    return F(F(F(d[0] * d[1]) + F(d[1] * d[2])) + F(d[2] * d[0])) * F(2.0)


def cost(top, bottom, n):  # get_cost: surface area * num_triangles (size_t -> float)
    return F(surface_area(top, bottom) * F(n))


class Node:
    def __init__(self, top, bottom, tris):
        self.top, self.bottom, self.tris = top, bottom, tris  # tris: indices into the input, in node order
        self.left = self.right = None


def find_object_split(n_tris, axis, bin_top, bin_bottom, bin_count):  # bvh.cpp:54-105
    num_bins = len(bin_count)
    best = (FLT_MAX, 0, -1, None, None)  # cost, split index, axis, left box, right box (make_default)
    right_top = np.empty_like(bin_top)
    right_bottom = np.empty_like(bin_bottom)
    right_top[-1], right_bottom[-1] = bin_top[-1], bin_bottom[-1]
    for i in range(num_bins - 2, -1, -1):
        right_top[i] = np.maximum(right_top[i + 1], bin_top[i])
        right_bottom[i] = np.minimum(right_bottom[i + 1], bin_bottom[i])
    left_top, left_bottom = bin_top[0].copy(), bin_bottom[0].copy()
    left_n = int(bin_count[0])
    for split in range(num_bins - 1):
        right_n = n_tris - left_n
        if left_n > 2 and right_n > 2:
            total = F(cost(left_top, left_bottom, left_n) + cost(right_top[split], right_bottom[split], right_n))
            other = (total, split, axis, (left_top.copy(), left_bottom.copy()),
                     (right_top[split].copy(), right_bottom[split].copy()))
            best = best if best[0] < other[0] else other  # SplitParams::min
        left_top = np.maximum(left_top, bin_top[split + 1])
        left_bottom = np.minimum(left_bottom, bin_bottom[split + 1])
        left_n += int(bin_count[split + 1])
    return best


def build_node(node, depth, tri_top, tri_bottom):  # bvh.cpp:136-203
    n = len(node.tris)
    if n <= MIN_TRIANGLES_PER_LEAF:
        return
    best = (cost(node.top, node.bottom, n), 0, -1, None, None)
    num_bins = int(MAX_BINS / F(depth + 1))
    if num_bins <= 1:
        return
    num_splits = num_bins - 1
    start, end = node.bottom, node.top
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        step = (end - start) / F(num_bins)
        inv = F(1.0) / step
        top, bottom = tri_top[node.tris], tri_bottom[node.tris]
        center = (top + bottom) / F(2.0)
        scaled = (center - start) * inv
    split_index = np.zeros((n, 3), np.int64)
    for axis in range(3):
        if abs(F(end[axis] - start[axis])) < F(1e-4):
            continue  # the indices of a skipped axis are never looked at
        split_index[:, axis] = np.minimum(scaled[:, axis].astype(np.int64), num_splits)  # uvec3(...) truncates
        bin_top = np.full((num_bins, 3), -FLT_MAX, np.float32)
        bin_bottom = np.full((num_bins, 3), FLT_MAX, np.float32)
        bin_count = np.zeros(num_bins, np.int64)
        np.maximum.at(bin_top, split_index[:, axis], top)
        np.minimum.at(bin_bottom, split_index[:, axis], bottom)
        np.add.at(bin_count, split_index[:, axis], 1)
        other = find_object_split(n, axis, bin_top, bin_bottom, bin_count)
        best = best if best[0] < other[0] else other
    if best[2] == -1:
        return
    goes_left = split_index[:, best[2]] <= best[1]  # bvh.cpp:120-128, in order
    node.left = Node(best[3][0], best[3][1], node.tris[goes_left])
    node.right = Node(best[4][0], best[4][1], node.tris[~goes_left])
    node.tris = None
    build_node(node.left, depth + 1, tri_top, tri_bottom)
    build_node(node.right, depth + 1, tri_top, tri_bottom)


def flatten(node, flat, order):  # bvh.cpp:211-249
    index = len(flat)
    rec = [node.top[0], node.top[1], node.top[2], F(0), node.bottom[0], node.bottom[1], node.bottom[2], F(0)]
    flat.append(rec)
    if node.left is None:
        rec[3] = F(len(order))
        rec[7] = -F(len(node.tris))
        order.extend(int(t) for t in node.tris)
    else:
        if surface_area(node.right.top, node.right.bottom) > surface_area(node.left.top, node.left.bottom):
            node.left, node.right = node.right, node.left
        flatten(node.left, flat, order)
        rec[7] = F(flatten(node.right, flat, order))
    return index


def build(verts):
    """verts float32 [n, 9] -> (flat nodes float32 [m, 8] = top.xyz, w, bottom.xyz, w; triangle order)."""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3, 3)
    tri_top, tri_bottom = v.max(axis=1), v.min(axis=1)  # Triangle::get_bounds
    root = Node(tri_top.max(axis=0), tri_bottom.min(axis=0), np.arange(len(v)))
    build_node(root, 0, tri_top, tri_bottom)
    flat, order = [], []
    flatten(root, flat, order)
    return np.array(flat, np.float32), np.array(order, np.uint32)


def glm_inverse(m):
    """glm::inverse(mat4), scalar path, in its operation order (include/glm/detail/func_matrix.inl:294-352).
    m[c][r] float32 (column c, row r) -> same layout."""
    c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3]
    c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3]
    c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3]
    c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3]
    c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3]
    c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3]
    c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2]
    c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2]
    c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2]
    c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3]
    c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3]
    c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3]
    c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2]
    c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2]
    c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2]
    c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1]
    c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1]
    c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1]
    v4 = lambda *a: np.array(a, np.float32)  # noqa: E731
    fac0, fac1, fac2 = v4(c00, c00, c02, c03), v4(c04, c04, c06, c07), v4(c08, c08, c10, c11)
    fac3, fac4, fac5 = v4(c12, c12, c14, c15), v4(c16, c16, c18, c19), v4(c20, c20, c22, c23)
    vec0, vec1 = v4(m[1][0], m[0][0], m[0][0], m[0][0]), v4(m[1][1], m[0][1], m[0][1], m[0][1])
    vec2, vec3 = v4(m[1][2], m[0][2], m[0][2], m[0][2]), v4(m[1][3], m[0][3], m[0][3], m[0][3])
    sign_a, sign_b = v4(1, -1, 1, -1), v4(-1, 1, -1, 1)
    inverse = np.stack([(vec1 * fac0 - vec2 * fac1 + vec3 * fac2) * sign_a,
                        (vec0 * fac0 - vec2 * fac3 + vec3 * fac4) * sign_b,
                        (vec0 * fac1 - vec1 * fac3 + vec3 * fac5) * sign_a,
                        (vec0 * fac2 - vec1 * fac4 + vec2 * fac5) * sign_b])
    dot0 = m[0] * inverse[:, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        one_over_determinant = F(1.0) / F(F(dot0[0] + dot0[1]) + F(dot0[2] + dot0[3]))
        return inverse * one_over_determinant


def woop(tri):
    """intersectable_manager.cpp:129-145 for one triangle (float32 [3, 3]) -> float32 [3, 4]."""
    v1, v2, v3 = (np.asarray(t, np.float32) for t in tri)
    e1, e2 = v2 - v1, v3 - v1
    cross = np.array([e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1]],
                     np.float32)  # glm::cross
    with np.errstate(divide="ignore", invalid="ignore"):
        length2 = F(F(cross[0] * cross[0] + cross[1] * cross[1]) + cross[2] * cross[2])  # glm::dot
        normal = cross * (F(1.0) / np.sqrt(length2))                                    # v * inversesqrt(dot(v, v))
        m = np.zeros((4, 4), np.float32)
        m[0, :3], m[1, :3], m[2, :3], m[3, :3] = e1, e2, normal - v1, v1
        m[3, 3] = 1.0
        inv = glm_inverse(m)
    return inv.T[:3].copy()  # transpose(inverse)[k] = row k of the inverse
