"""Reader for Circom's binary `.r1cs` format — what /root/reference/legogroth16/src/circom/r1cs_reader.rs:16-140
parses: magic "r1cs", version 1, sections {1: header, 2: constraints, 3: wire2label}; 32-byte little-endian field elements;
a constraint is three linear combinations (A, B, C) of (wire id, coefficient) terms with  <A,w> * <B,w> = <C,w>.
Wire order is Circom's: 0 = one, public outputs, public inputs, private inputs, intermediates — i.e. the assignment order
(1, instance..., witness...) that `dgpu_witness_map` / `DeviceR1cs` take (legogroth16/src/circom/circuit.rs:85-141)."""
import struct
import numpy as np

BLS12_381_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BN128_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class R1csFile:
    def __init__(self, data):
        try:
            self._parse(data)
        except struct.error:                       # a truncated file is a parsing error like any other (never a struct.error for the caller)
            raise ValueError("unexpected end of file") from None

    def _parse(self, data):
        if len(data) < 12:
            raise ValueError("unexpected end of file")
        if data[:4] != b"r1cs":
            raise ValueError("Invalid magic number")
        version, nsec = struct.unpack_from("<II", data, 4)
        if version != 1:
            raise ValueError("Unsupported version")
        off, secs = 12, {}
        for _ in range(nsec):
            typ, size = struct.unpack_from("<IQ", data, off)
            off += 12
            secs[typ] = (off, size)
            off += size
        if 1 not in secs or 2 not in secs:
            raise ValueError("missing header or constraint section")
        h, hsize = secs[1]
        (fs,) = struct.unpack_from("<I", data, h)
        if fs != 32 or hsize != 32 + fs:
            raise ValueError("This parser only supports 32-byte fields")
        self.prime = int.from_bytes(data[h + 4:h + 36], "little")
        self.n_wires, self.n_pub_out, self.n_pub_in, self.n_prv_in = struct.unpack_from("<IIII", data, h + 36)
        (self.n_labels,) = struct.unpack_from("<Q", data, h + 52)
        (self.n_constraints,) = struct.unpack_from("<I", data, h + 60)
        p = secs[2][0]
        self.constraints = []
        for _ in range(self.n_constraints):
            lcs = []
            for _ in range(3):
                (nt,) = struct.unpack_from("<I", data, p); p += 4
                terms = []
                for _ in range(nt):
                    (wire,) = struct.unpack_from("<I", data, p)
                    terms.append((int.from_bytes(data[p + 4:p + 36], "little"), wire))
                    p += 36
                lcs.append(terms)
            self.constraints.append(tuple(lcs))
        # section 3, wire2label (r1cs_reader.rs:91-101, read_map :219-238): n_wires u64 labels, wire 0 -> label 0
        # (the reference looks the section up unconditionally, :91-96: a file without it is an R1CSFileParsing error there, so it is one here)
        if 3 not in secs:
            raise ValueError("No section offset for wire2label type found")
        w, wsize = secs[3]
        if wsize != self.n_wires * 8:
            raise ValueError("Invalid map section size")
        self.wire_mapping = list(struct.unpack_from("<%dQ" % self.n_wires, data, w))
        if self.wire_mapping and self.wire_mapping[0] != 0:
            raise ValueError("Wire 0 should always be mapped to 0")

    @property
    def curve(self):
        """the curve whose scalar field the file was compiled for (r1cs_reader.rs:186-200: anything else is IncompatibleWithCurve)"""
        return {BLS12_381_ORDER: "bls12_381", BN128_ORDER: "bn128"}.get(self.prime)

    @classmethod
    def from_path(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    @property
    def num_inputs(self):
        """instance variables incl. the constant one (arkworks `num_instance_variables`)"""
        return 1 + self.n_pub_out + self.n_pub_in

    def rows(self, k):
        return [c[k] for c in self.constraints]

    def csr(self):
        from .qap import csr
        return [csr(self.rows(k)) for k in range(3)]

    def is_satisfied(self, w):
        r = self.prime
        dot = lambda lc: sum(co * w[i] for co, i in lc) % r
        return all(dot(a) * dot(b) % r == dot(c) for a, b, c in self.constraints)
