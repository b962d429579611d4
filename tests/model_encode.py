"""Array-level model of the device SPSS encode (test infrastructure).

This is the data-parallel formulation that csrc/ksh_encode.hip implements, written
with the same arrays and the same rules, in plain Python/numpy so that it can be
checked against the oracle (oracle/ko_spss.h, the restatement of
lib/core/spss.h:230-615 and :1039-1858) on the CPU.  It is NOT the oracle and not
product code.

Vocabulary.  A k-mer is its index t in the set's ascending order.  A *state*
s = 2 t + d walks k-mer t forward (d = 0: enters through its left side, leaves
through its right side, spelled as is) or reversed (d = 1: enters right, leaves
left, spelled reverse-complemented).  A side has a *link* when it has exactly one
neighbour whose facing side also has exactly one neighbour (spss.h:276-313).
succ(2 t + d) = 2 y + (d xor same_side) over the link of the leaving side.
"""
import numpy as np

NONE = -1


def revcomp1(x, k):
    out = 0
    for _ in range(k):
        out = (out << 2) | (3 - (x & 3))
        x >>= 2
    return out


class EncodeModel:
    def __init__(self, kmers, k):
        self.k = k
        self.km = [int(v) for v in kmers]
        self.n = len(self.km)
        self.index = {v: i for i, v in enumerate(self.km)}
        self.mask = (1 << (2 * k)) - 1

    # ---- k-mer helpers --------------------------------------------------------
    def canon(self, x):
        return min(x, revcomp1(x, self.k))

    def oriented(self, t, d):
        return self.km[t] if d == 0 else revcomp1(self.km[t], self.k)

    # neighbours of one side of k-mer t, in the reference's enumeration order
    # (c = A, C, G, T; spss.h:238-273).  Returns [(c, index, same_side)].
    def side_neighbours(self, t, side):
        x = self.km[t]
        out = []
        for c in range(4):
            if side == "R":
                y = ((x << 2) & self.mask) | c
            else:
                y = (x >> 2) | (c << (2 * (self.k - 1)))
            z = self.canon(y)
            if z == x:
                continue
            i = self.index.get(z)
            if i is None:
                continue
            out.append((c, i, z != y))
        return out

    # ---- E1: neighbour counts and links -----------------------------------------
    def build_links(self):
        n = self.n
        self.cnt = {"L": [0] * n, "R": [0] * n}
        self.single = {"L": [None] * n, "R": [None] * n}
        for t in range(n):
            for side in "LR":
                nb = self.side_neighbours(t, side)
                self.cnt[side][t] = len(nb)
                if len(nb) == 1:
                    self.single[side][t] = (nb[0][1], nb[0][2])
        self.link = {"L": [None] * n, "R": [None] * n}
        for t in range(n):
            for side in "LR":
                s = self.single[side][t]
                if s is None:
                    continue
                y, same = s
                facing = side if same else ("L" if side == "R" else "R")
                if self.cnt[facing][y] == 1:
                    self.link[side][t] = s

    def succ(self, s):
        t, d = s >> 1, s & 1
        lk = self.link["R" if d == 0 else "L"][t]
        if lk is None:
            return NONE
        y, same = lk
        return 2 * y + (d ^ int(same))

    def has_pred(self, s):
        t, d = s >> 1, s & 1
        return self.link["L" if d == 0 else "R"][t] is not None

    # ---- E2/E3: unitigs ----------------------------------------------------------------
    def build_unitigs(self):
        n = self.n
        # walks from every chain start; info[s] = (start state, position)
        info = [None] * (2 * n)
        for s0 in range(2 * n):
            if self.has_pred(s0):
                continue
            s, p = s0, 0
            while s != NONE:
                info[s] = (s0, p)
                p += 1
                s = self.succ(s)
        # choose, per k-mer, the chain that starts at the larger end (spss.h:511,555)
        self.head = [NONE] * n      # start k-mer of the unitig
        self.pos = [0] * n          # position in the unitig, in k-mers
        self.ori = [0] * n          # 0 spelled as is, 1 reverse-complemented
        self.ulen = {}              # head -> number of k-mers
        self.uclass = {}            # head -> 0 single, 1 from a left terminal, 2 from a right terminal, 3 loop
        covered = [False] * n
        for t in range(n):
            i0, i1 = info[2 * t], info[2 * t + 1]
            if i0 is None:
                continue            # on a loop
            assert i1 is not None
            st0, st1 = i0[0] >> 1, i1[0] >> 1   # chain of (t,0) starts at st0 and ends at st1
            d = 0 if st0 >= st1 else 1
            s0, p = (i0 if d == 0 else i1)
            self.head[t], self.pos[t], self.ori[t] = s0 >> 1, p, d
            covered[t] = True
            h = s0 >> 1
            self.ulen[h] = self.ulen.get(h, 0) + 1
            if p == 0:
                length_other = (i1 if d == 0 else i0)[1]
                single = length_other == 0 and (st0 == st1)
                self.uclass[h] = 0 if single else (1 if (s0 & 1) == 0 else 2)
        # loops: every uncovered k-mer with d = 0 walks until it meets a smaller k-mer or closes
        for t in range(n):
            if covered[t]:
                continue
            s, p, is_min = 2 * t, 0, True
            visit = []
            while True:
                visit.append((s, p))
                s = self.succ(s)
                p += 1
                assert s != NONE
                if (s >> 1) < t:
                    is_min = False
                    break
                if s == 2 * t:
                    break
                assert (s >> 1) != t, "a loop passes a k-mer once"
            if not is_min:
                continue
            for (s, p) in visit:
                self.head[s >> 1], self.pos[s >> 1], self.ori[s >> 1] = t, p, s & 1
            self.ulen[t] = len(visit)
            self.uclass[t] = 3
        # unitig order: (class, head k-mer) ascending == the reference's push order at n_workers == 1
        heads = sorted(self.ulen, key=lambda h: (self.uclass[h], h))
        self.uid_of_head = {h: i for i, h in enumerate(heads)}
        self.n_unitigs = len(heads)
        self.u_head = heads
        self.u_len = [self.ulen[h] for h in heads]
        # members by position
        self.u_members = [[None] * self.ulen[h] for h in heads]
        for t in range(n):
            u = self.uid_of_head[self.head[t]]
            self.u_members[u][self.pos[t]] = (t, self.ori[t])

    def unitig_string(self, u):
        members = self.u_members[u]
        s = self._str(self.oriented(*members[0]))
        for (t, d) in members[1:]:
            s += "ACGT"[self.oriented(t, d) & 3]
        return s

    def _str(self, x):
        return "".join("ACGT"[(x >> (2 * (self.k - 1 - i))) & 3] for i in range(self.k))

    # ---- E4: unitig graph ------------------------------------------------------------------
    # vertex v = 2 u + side (side 0 = left, 1 = right).  edges[v][c] = other vertex or NONE.
    def build_edges(self):
        U = self.n_unitigs
        self.edges = [[NONE] * 4 for _ in range(2 * U)]
        for u in range(U):
            members = self.u_members[u]
            for side in (0, 1):
                t, d = members[0] if side == 0 else members[-1]
                o = self.oriented(t, d)
                for c in range(4):
                    if side == 1:
                        y = ((o << 2) & self.mask) | c
                    else:
                        y = (o >> 2) | (c << (2 * (self.k - 1)))
                    z = self.canon(y)
                    i = self.index.get(z)
                    if i is None:
                        continue
                    u2 = self.uid_of_head[self.head[i]]
                    if u2 == u:
                        continue
                    # side of k-mer z that this edge touches
                    if side == 1:
                        f = "L" if y == z else "R"
                    else:
                        f = "R" if y == z else "L"
                    first, last = self.u_members[u2][0], self.u_members[u2][-1]
                    enter = "L" if first[1] == 0 else "R"
                    leave = "R" if last[1] == 0 else "L"
                    if first[0] == i and enter == f:
                        side2 = 0
                    else:
                        assert last[0] == i and leave == f, "neighbour of a unitig end is a unitig end"
                        side2 = 1
                    self.edges[2 * u + side][c] = 2 * u2 + side2

    # priority of slot (v, c) in the reference's sequential sweep (spss.h:1450-1493):
    # unitig i in ascending order, right edges before left edges, c ascending.
    @staticmethod
    def slot_priority(v, c):
        u, side = v >> 1, v & 1
        return u * 8 + (0 if side == 1 else 4) + c

    def greedy_sequential(self):
        U = self.n_unitigs
        mate = [NONE] * (2 * U)
        for u in range(U):
            for side in (1, 0):
                v = 2 * u + side
                for c in range(4):
                    w = self.edges[v][c]
                    if w == NONE:
                        continue
                    if mate[v] == NONE and mate[w] == NONE:
                        mate[v], mate[w] = w, v
        return mate

    def greedy_rounds(self):
        """Lexicographically-first maximal matching by rounds of mutual minima."""
        U = self.n_unitigs
        # edge priority = the earlier of its two slots
        prio = [[None] * 4 for _ in range(2 * U)]
        for v in range(2 * U):
            for c in range(4):
                w = self.edges[v][c]
                if w == NONE:
                    continue
                back = [c2 for c2 in range(4) if self.edges[w][c2] == v]
                assert len(back) == 1, "edges are symmetric and simple"
                prio[v][c] = min(self.slot_priority(v, c), self.slot_priority(w, back[0]))
        mate = [NONE] * (2 * U)
        rounds = 0
        while True:
            best = [None] * (2 * U)
            for v in range(2 * U):
                if mate[v] != NONE:
                    continue
                for c in range(4):
                    w = self.edges[v][c]
                    if w == NONE or mate[w] != NONE:
                        continue
                    if best[v] is None or prio[v][c] < best[v][0]:
                        best[v] = (prio[v][c], w)
            any_live = False
            new = []
            for v in range(2 * U):
                if best[v] is None:
                    continue
                any_live = True
                w = best[v][1]
                if best[w] is not None and best[w][1] == v and best[w][0] == best[v][0]:
                    new.append((v, w))
            for v, w in new:
                mate[v] = w
            rounds += 1
            if not any_live:
                break
        self.rounds = rounds
        return mate

    # ---- E5: loops of the path cover (spss.h:1541-1647) ----------------------------------------
    def cut_loops(self, mate):
        U = self.n_unitigs
        mate = list(mate)
        # components by walking; a component is a loop iff no vertex in it is free
        comp = [NONE] * U
        for u in range(U):
            if comp[u] != NONE:
                continue
            stack, members = [u], []
            comp[u] = u
            while stack:
                a = stack.pop()
                members.append(a)
                for side in (0, 1):
                    w = mate[2 * a + side]
                    if w != NONE and comp[w >> 1] == NONE:
                        comp[w >> 1] = u
                        stack.append(w >> 1)
            if any(mate[2 * a] == NONE or mate[2 * a + 1] == NONE for a in members):
                continue
            # the reference's sequential union-by-rank decides which node loses its left edge
            nodes = sorted(members)
            parent = {a: a for a in nodes}
            rank = {a: 0 for a in nodes}

            def find(a):
                while parent[a] != a:
                    a = parent[a]
                return a

            def unite(a, b):
                a, b = find(a), find(b)
                if a == b:
                    return
                if rank[a] > rank[b] or (rank[a] == rank[b] and a > b):
                    a, b = b, a
                parent[a] = b
                if rank[a] == rank[b]:
                    rank[b] += 1

            for a in nodes:
                unite(a, mate[2 * a] >> 1)       # edge_left first (spss.h:1553-1557)
                unite(a, mate[2 * a + 1] >> 1)
            root = find(nodes[0])
            w = mate[2 * root]
            mate[2 * root] = NONE
            mate[w] = NONE
        return mate

    # ---- E6: stitch (spss.h:1649-1829) -------------------------------------------------------------
    def stitch(self, mate):
        U = self.n_unitigs
        left_t, right_t, both_t = [], [], []
        for u in range(U):
            hl, hr = mate[2 * u] != NONE, mate[2 * u + 1] != NONE
            if not hl and not hr:
                both_t.append(u)
            elif not hl:
                left_t.append(u)
            elif not hr:
                right_t.append(u)

        def walk(u, going_right):
            path = []
            while True:
                path.append((u, not going_right))   # (unitig, complemented)
                w = mate[2 * u + (1 if going_right else 0)]
                if w == NONE:
                    return path
                u, entered = w >> 1, w & 1
                going_right = entered == 0          # entered through its left side -> keeps going right

        def string_of(path):
            s = ""
            for idx, (u, comp) in enumerate(path):
                us = self.unitig_string(u)
                if comp:
                    us = us[::-1].translate(str.maketrans("ACGT", "TGCA"))
                s += us if idx == 0 else us[self.k - 1:]
            return s

        out = []
        for u in left_t:
            p = walk(u, True)
            if p[0][0] > p[-1][0]:
                continue
            out.append(string_of(p))
        for u in right_t:
            p = walk(u, False)
            if p[0][0] > p[-1][0]:
                continue
            out.append(string_of(p))
        for u in both_t:
            out.append(self.unitig_string(u))
        return out

    # ---- whole encode ----------------------------------------------------------------------------------
    def unitigs(self):
        self.build_links()
        self.build_unitigs()
        return [self.unitig_string(u) for u in range(self.n_unitigs)]

    def spss(self):
        self.build_links()
        self.build_unitigs()
        self.build_edges()
        mate = self.greedy_rounds()
        assert mate == self.greedy_sequential()
        mate = self.cut_loops(mate)
        return self.stitch(mate)
