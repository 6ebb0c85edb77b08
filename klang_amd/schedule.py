"""Execution order of a block network.

Host-side restatement of klang/execution.py:8-64 and klang/graph.py:11-175:
breadth-first discovery of the network from the given blocks, removal of back
edges found by a depth-first colouring, then the reference's queue-driven
topological sort.  The traversal rules are kept exactly (successors ascending,
ready nodes push their successors to the FRONT of the queue) because the
resulting orders are part of the observable behaviour: tests/test_schedule.py
holds the 13 golden orders of the reference's tests/test_graph.py:140-243.
Graphs are adjacency lists here, not dense matrices.
"""
import collections

from klang_amd.block import input_neighbors, output_neighbors


def traverse_network(blocks):
    """All blocks reachable from `blocks` (in discovery order) and the edges."""
    seen, known, edges = [], set(), set()   # `known`: O(1) membership for large patches
    todo = collections.deque(blocks)
    while todo:
        blk = todo.popleft()
        if blk in known:
            continue
        known.add(blk)
        seen.append(blk)
        for nxt in output_neighbors(blk):
            edges.add((blk, nxt))
            todo.append(nxt)
        for prv in input_neighbors(blk):
            edges.add((prv, blk))
            todo.append(prv)
    return seen, edges


def _adjacency(n_nodes, edges):
    succ = [set() for _ in range(n_nodes)]
    for a, b in edges:
        succ[a].add(b)
    # (most nodes of a large patch have one successor: nothing to sort)
    return [sorted(s) if len(s) > 1 else list(s) for s in succ]


def find_back_edges(n_nodes, edges):
    """Edges that close a cycle in a depth-first walk started from the lowest
    unvisited node, successors in ascending order (graph.py:91-128)."""
    succ = _adjacency(n_nodes, edges)
    WHITE, GRAY, BLACK = 0, 1, 2
    colour = [WHITE] * n_nodes
    back = []

    def visit(node):
        colour[node] = GRAY
        for child in succ[node]:
            if colour[child] == BLACK:
                continue
            if colour[child] == GRAY:
                back.append((node, child))
                continue
            visit(child)
        colour[node] = BLACK

    for node in range(n_nodes):
        if colour[node] == WHITE:
            visit(node)
    return back


def topological_order(n_nodes, edges):
    """Execution order of nodes 0..n-1 for a directed graph that may have cycles
    (graph.py:151-175)."""
    edges = set(edges) - set(find_back_edges(n_nodes, edges))
    succ = _adjacency(n_nodes, edges)
    pred = [[] for _ in range(n_nodes)]
    for a, b in edges:
        pred[b].append(a)
    queue = collections.deque(range(n_nodes))
    order, done = [], set()
    while queue:
        node = queue.popleft()
        if node in done:
            continue
        if all(p in done for p in pred[node]):
            queue.extendleft(succ[node])
            order.append(node)
            done.add(node)
        else:
            queue.extend(succ[node])
    return order


def determine_execution_order(blocks):
    """execution.py:37-58."""
    vertices, edges = traverse_network(list(blocks))
    if not edges:
        return vertices
    index = {blk: i for i, blk in enumerate(vertices)}
    order = topological_order(len(vertices), [(index[a], index[b]) for a, b in edges])
    return [vertices[i] for i in order]


def execute(blocks):
    for blk in blocks:
        blk.update()
