"""The MiMC `LongsightF322p3` preimage circuit of the reference's integration test (legogroth16/tests/mimc.rs:35-147), restated as an R1CS
in the row format of lego_setup.circuit: 322 rounds of  xL, xR := xR + (xL + C_i)^3, xL ; two constraints per round,
    (xL + C_i) * (xL + C_i) = tmp          tmp * (xL + C_i) = new_xL - xR
variables: 0 = one, 1 = image (the last new_xL, public); witnesses in allocation order xl, xr, then per round tmp, new_xL."""

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
MIMC_ROUNDS = 322


def mimc(xl, xr, constants):
    assert len(constants) == MIMC_ROUNDS
    for c in constants:
        t = (xl + c) % R
        xl, xr = (pow(t, 3, R) + xr) % R, xl
    return xl


def circuit(xl, xr, constants):
    n_inst = 2
    wit = [xl % R, xr % R]                      # witness values in allocation order
    A, B, C = [], [], []
    w = lambda k: n_inst + k                     # z index of witness k
    xl_var, xr_var, xl_val, xr_val = w(0), w(1), xl % R, xr % R
    image = None
    for i, c in enumerate(constants):
        base = (xl_val + c) % R
        tmp_val = base * base % R
        wit.append(tmp_val); tmp_var = w(len(wit) - 1)
        lin = [(1, xl_var), (c % R, 0)]
        A.append(list(lin)); B.append(list(lin)); C.append([(1, tmp_var)])
        new_val = (base * tmp_val + xr_val) % R
        if i == len(constants) - 1:
            image, new_var = new_val, 1          # the last round's new_xL is the public input
        else:
            wit.append(new_val); new_var = w(len(wit) - 1)
        A.append([(1, tmp_var)]); B.append(list(lin)); C.append([(1, new_var), (R - 1, xr_var)])
        xr_var, xr_val = xl_var, xl_val
        xl_var, xl_val = new_var, new_val
    z = [1, image] + wit
    return {"A": A, "B": B, "C": C, "z": z, "n_inst": n_inst, "n_wit": len(wit), "n_cons": len(A)}
