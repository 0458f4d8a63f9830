"""Builtin message / reduce descriptors of the DGL shim (test infrastructure only)."""

import operator


class _Binary:
    def __init__(self, lhs_target, rhs_target, op, lhs, rhs, out):
        self.lhs_target, self.rhs_target, self.op = lhs_target, rhs_target, op
        self.lhs, self.rhs, self.out = lhs, rhs, out


class _CopyE:
    def __init__(self, name, out):
        self.name, self.out = name, out


class _CopyU:
    def __init__(self, name, out):
        self.name, self.out = name, out


class _Sum:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def u_add_v(lhs, rhs, out):
    return _Binary("u", "v", operator.add, lhs, rhs, out)


def v_sub_u(lhs, rhs, out):
    # DGL: out = v[lhs] - u[rhs]
    return _Binary("v", "u", operator.sub, lhs, rhs, out)


def u_mul_e(lhs, rhs, out):
    return _Binary("u", "e", operator.mul, lhs, rhs, out)


def copy_e(name, out):
    return _CopyE(name, out)


def copy_u(name, out):
    return _CopyU(name, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return _Sum(msg, out)
