"""The photon-tree vectors the reference's own tests hold (raytracer/test/KDTree.cpp:20-163 and
test/KDTreeBuilder.cpp:20-38): hand-built trees of 3-4 points (node i holds point i; children are node indices or -1;
axis 0 = x, 1 = y), a target, the number of neighbours asked for, and the expected result order.

`test_max_dist` is the value the reference test asserts for the returned radius, as the index of the point whose
distance to the target it names.  For the three two-neighbour cases (1, 2, 6) the test names the NEAREST result, while
KDTreeNode::insert (photonmapping/KDTree.h:230-250) returns the distance of the LAST entry of the full result list,
i.e. the farthest of the two: those assertions cannot hold against the reference's own source (the tests are not part
of its default build, raytracer/CMakeLists.txt:39-41), so they are recorded here and checked as "source says",
not as expectations.  Order and count assertions hold for all cases.
"""
CASES = [
    # name, points, child0, child1, axis, target, count, expected order, point named by the test's maxDist assertion
    ("NNearestElements1", [(0, 0, 0), (-5, 0, 0), (5, 0, 0)], [1, -1, -1], [2, -1, -1], [0, 0, 0], (-4, 0, 0), 2, [1, 0], 1),
    ("NNearestElements2", [(0, 5, 0), (-5, 0, 0), (5, 0, 0)], [1, -1, -1], [2, -1, -1], [0, 0, 0], (5, 4, 0), 2, [2, 0], 2),
    ("NNearestElements3", [(0, 0, 0), (-1, 4, 0), (5, 0, 0)], [1, -1, -1], [2, -1, -1], [0, 0, 0], (1, 4, 0), 1, [1], 1),
    ("NNearestElements4", [(0, 0, 0), (-5, 0, 0), (2, 0, 0), (2, 1, 0)], [1, -1, -1, -1], [2, -1, 3, -1], [0, 0, 1, 0], (2, -1, 0), 1, [2], 2),
    ("NNearestElements5", [(0, 0, 0), (-5, 0, 0), (2, 0, 0), (2, 1, 0)], [1, -1, -1, -1], [2, -1, 3, -1], [0, 0, 1, 0], (-1, 1, 0), 1, [0], 0),
    ("NNearestElements6", [(0, 0, 0), (-5, 0, 0), (3, 0, 0), (1, 1, 0)], [1, -1, -1, -1], [2, -1, 3, -1], [0, 0, 1, 0], (3, 1, 0), 2, [2, 3], 2),
    ("KDTreeBuilder", [(0, 0, 0), (-5, 0, 0), (5, 0, 0)], [1, -1, -1], [2, -1, -1], [0, 0, 0], (-4, 0, 0), 2, [1, 0], None),
]
