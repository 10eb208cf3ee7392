"""Parameter sets of the reference, as plain data.

shortint: ``concrete-shortint/src/parameters/mod.rs:28-45`` (struct) and ``:161-808`` (the sets);
boolean: ``concrete-boolean/src/parameters/mod.rs:83-111``.
Field names and meanings are the reference's.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Parameters:
    """shortint parameter set (64-bit torus)."""
    lwe_dimension: int
    glwe_dimension: int
    polynomial_size: int
    lwe_modular_std_dev: float
    glwe_modular_std_dev: float
    pbs_base_log: int
    pbs_level: int
    ks_base_log: int
    ks_level: int
    message_modulus: int
    carry_modulus: int

    torus_bits = 64


@dataclass(frozen=True)
class BooleanParameters:
    """boolean parameter set (32-bit torus)."""
    lwe_dimension: int
    glwe_dimension: int
    polynomial_size: int
    lwe_modular_std_dev: float
    glwe_modular_std_dev: float
    pbs_base_log: int
    pbs_level: int
    ks_base_log: int
    ks_level: int

    torus_bits = 32
    message_modulus = 0
    carry_modulus = 0


PARAM_MESSAGE_1_CARRY_0 = Parameters(567, 5, 256, 0.00017395369678340785, 3.7411618952047216e-10, 15, 1, 3, 3, 2, 1)  # parameters/mod.rs:161
PARAM_MESSAGE_1_CARRY_1 = Parameters(653, 6, 256, 3.604499526942373e-05, 3.4525330484572114e-12, 18, 1, 4, 3, 2, 2)  # parameters/mod.rs:179
PARAM_MESSAGE_2_CARRY_0 = Parameters(653, 5, 256, 3.604499526942373e-05, 3.7411618952047216e-10, 15, 1, 4, 3, 4, 1)  # parameters/mod.rs:197
PARAM_MESSAGE_1_CARRY_2 = Parameters(698, 3, 512, 1.5818059929193196e-05, 3.4525330484572114e-12, 18, 1, 3, 4, 2, 4)  # parameters/mod.rs:215
PARAM_MESSAGE_2_CARRY_1 = Parameters(696, 3, 512, 1.640781036519474e-05, 3.4525330484572114e-12, 18, 1, 4, 3, 4, 2)  # parameters/mod.rs:233
PARAM_MESSAGE_3_CARRY_0 = Parameters(694, 3, 512, 1.701954867950249e-05, 3.4525330484572114e-12, 18, 1, 4, 3, 8, 1)  # parameters/mod.rs:251
PARAM_MESSAGE_1_CARRY_3 = Parameters(771, 2, 1024, 4.158126532841584e-06, 2.9403601535432533e-16, 23, 1, 4, 3, 2, 8)  # parameters/mod.rs:269
PARAM_MESSAGE_2_CARRY_2 = Parameters(769, 2, 1024, 4.3131554647504185e-06, 2.9403601535432533e-16, 23, 1, 4, 3, 4, 4)  # parameters/mod.rs:287
PARAM_MESSAGE_3_CARRY_1 = Parameters(769, 2, 1024, 4.3131554647504185e-06, 2.9403601535432533e-16, 23, 1, 4, 3, 8, 2)  # parameters/mod.rs:305
PARAM_MESSAGE_4_CARRY_0 = Parameters(769, 2, 1024, 4.3131554647504185e-06, 2.9403601535432533e-16, 23, 1, 4, 3, 16, 1)  # parameters/mod.rs:323
PARAM_MESSAGE_1_CARRY_4 = Parameters(769, 1, 2048, 4.3131554647504185e-06, 2.9403601535432533e-16, 23, 1, 3, 5, 2, 16)  # parameters/mod.rs:341
PARAM_MESSAGE_2_CARRY_3 = Parameters(755, 1, 2048, 5.572845359330198e-06, 2.9403601535432533e-16, 23, 1, 3, 5, 4, 8)  # parameters/mod.rs:359
PARAM_MESSAGE_3_CARRY_2 = Parameters(754, 1, 2048, 5.6757818866051225e-06, 2.9403601535432533e-16, 23, 1, 3, 5, 8, 4)  # parameters/mod.rs:377
PARAM_MESSAGE_4_CARRY_1 = Parameters(754, 1, 2048, 5.6757818866051225e-06, 2.9403601535432533e-16, 23, 1, 3, 5, 16, 2)  # parameters/mod.rs:395
PARAM_MESSAGE_5_CARRY_0 = Parameters(754, 1, 2048, 5.6757818866051225e-06, 2.9403601535432533e-16, 23, 1, 3, 5, 32, 1)  # parameters/mod.rs:413
PARAM_MESSAGE_1_CARRY_5 = Parameters(824, 1, 4096, 1.5762180593038625e-06, 2.168404344971009e-19, 15, 2, 3, 5, 2, 32)  # parameters/mod.rs:431
PARAM_MESSAGE_2_CARRY_4 = Parameters(824, 1, 4096, 1.5762180593038625e-06, 2.168404344971009e-19, 15, 2, 3, 5, 4, 16)  # parameters/mod.rs:449
PARAM_MESSAGE_3_CARRY_3 = Parameters(873, 1, 4096, 6.428797112843789e-07, 2.168404344971009e-19, 23, 1, 4, 4, 8, 8)  # parameters/mod.rs:467
PARAM_MESSAGE_4_CARRY_2 = Parameters(850, 1, 4096, 9.793771134612522e-07, 2.168404344971009e-19, 23, 1, 4, 4, 16, 4)  # parameters/mod.rs:485
PARAM_MESSAGE_5_CARRY_1 = Parameters(848, 1, 4096, 1.0158915837729808e-06, 2.168404344971009e-19, 23, 1, 4, 4, 32, 2)  # parameters/mod.rs:503
PARAM_MESSAGE_6_CARRY_0 = Parameters(847, 1, 4096, 1.0346562084806e-06, 2.168404344971009e-19, 23, 1, 4, 4, 64, 1)  # parameters/mod.rs:521
PARAM_MESSAGE_1_CARRY_6 = Parameters(879, 1, 8192, 5.760198979100338e-07, 2.168404344971009e-19, 15, 2, 3, 6, 2, 64)  # parameters/mod.rs:539
PARAM_MESSAGE_2_CARRY_5 = Parameters(878, 1, 8192, 5.866596131917157e-07, 2.168404344971009e-19, 15, 2, 3, 6, 4, 32)  # parameters/mod.rs:557
PARAM_MESSAGE_3_CARRY_4 = Parameters(877, 1, 8192, 5.974958556101962e-07, 2.168404344971009e-19, 15, 2, 3, 6, 8, 16)  # parameters/mod.rs:575
PARAM_MESSAGE_4_CARRY_3 = Parameters(877, 1, 8192, 5.974958556101962e-07, 2.168404344971009e-19, 15, 2, 3, 6, 16, 8)  # parameters/mod.rs:593
PARAM_MESSAGE_5_CARRY_2 = Parameters(877, 1, 8192, 5.974958556101962e-07, 2.168404344971009e-19, 15, 2, 3, 6, 32, 4)  # parameters/mod.rs:611
PARAM_MESSAGE_6_CARRY_1 = Parameters(893, 1, 8192, 4.458159540199854e-07, 2.168404344971009e-19, 23, 1, 3, 6, 64, 2)  # parameters/mod.rs:629
PARAM_MESSAGE_7_CARRY_0 = Parameters(880, 1, 8192, 5.655731455300567e-07, 2.168404344971009e-19, 23, 1, 3, 6, 128, 1)  # parameters/mod.rs:647
PARAM_MESSAGE_1_CARRY_7 = Parameters(954, 1, 16384, 1.4597704188641654e-07, 2.168404344971009e-19, 11, 3, 3, 6, 2, 128)  # parameters/mod.rs:665
PARAM_MESSAGE_2_CARRY_6 = Parameters(991, 1, 16384, 7.416217327637463e-08, 2.168404344971009e-19, 15, 2, 4, 5, 4, 64)  # parameters/mod.rs:683
PARAM_MESSAGE_3_CARRY_5 = Parameters(973, 1, 16384, 1.0310011520238092e-07, 2.168404344971009e-19, 15, 2, 4, 5, 8, 32)  # parameters/mod.rs:701
PARAM_MESSAGE_4_CARRY_4 = Parameters(953, 1, 16384, 1.486733969411098e-07, 2.168404344971009e-19, 15, 2, 3, 6, 16, 16)  # parameters/mod.rs:719
PARAM_MESSAGE_5_CARRY_3 = Parameters(952, 1, 16384, 1.5141955661224835e-07, 2.168404344971009e-19, 15, 2, 3, 6, 32, 8)  # parameters/mod.rs:737
PARAM_MESSAGE_6_CARRY_2 = Parameters(952, 1, 16384, 1.5141955661224835e-07, 2.168404344971009e-19, 15, 2, 3, 6, 64, 4)  # parameters/mod.rs:755
PARAM_MESSAGE_7_CARRY_1 = Parameters(952, 1, 16384, 1.5141955661224835e-07, 2.168404344971009e-19, 15, 2, 3, 6, 128, 2)  # parameters/mod.rs:773
PARAM_MESSAGE_8_CARRY_0 = Parameters(952, 1, 16384, 1.5141955661224835e-07, 2.168404344971009e-19, 15, 2, 3, 6, 256, 1)  # parameters/mod.rs:791

ALL_PARAMETER_VEC = (
    PARAM_MESSAGE_1_CARRY_1,
    PARAM_MESSAGE_1_CARRY_2,
    PARAM_MESSAGE_1_CARRY_3,
    PARAM_MESSAGE_1_CARRY_4,
    PARAM_MESSAGE_1_CARRY_5,
    PARAM_MESSAGE_1_CARRY_6,
    PARAM_MESSAGE_1_CARRY_7,
    PARAM_MESSAGE_2_CARRY_1,
    PARAM_MESSAGE_2_CARRY_2,
    PARAM_MESSAGE_2_CARRY_3,
    PARAM_MESSAGE_2_CARRY_4,
    PARAM_MESSAGE_2_CARRY_5,
    PARAM_MESSAGE_2_CARRY_6,
    PARAM_MESSAGE_3_CARRY_1,
    PARAM_MESSAGE_3_CARRY_2,
    PARAM_MESSAGE_3_CARRY_3,
    PARAM_MESSAGE_3_CARRY_4,
    PARAM_MESSAGE_3_CARRY_5,
    PARAM_MESSAGE_4_CARRY_1,
    PARAM_MESSAGE_4_CARRY_2,
    PARAM_MESSAGE_4_CARRY_3,
    PARAM_MESSAGE_4_CARRY_4,
    PARAM_MESSAGE_5_CARRY_1,
    PARAM_MESSAGE_5_CARRY_2,
    PARAM_MESSAGE_5_CARRY_3,
    PARAM_MESSAGE_6_CARRY_1,
    PARAM_MESSAGE_6_CARRY_2,
    PARAM_MESSAGE_7_CARRY_1,
)

BIVARIATE_PBS_COMPLIANT_PARAMETER_SET_VEC = (
    PARAM_MESSAGE_1_CARRY_1,
    PARAM_MESSAGE_1_CARRY_2,
    PARAM_MESSAGE_1_CARRY_3,
    PARAM_MESSAGE_1_CARRY_4,
    PARAM_MESSAGE_1_CARRY_5,
    PARAM_MESSAGE_1_CARRY_6,
    PARAM_MESSAGE_1_CARRY_7,
    PARAM_MESSAGE_2_CARRY_2,
    PARAM_MESSAGE_2_CARRY_3,
    PARAM_MESSAGE_2_CARRY_4,
    PARAM_MESSAGE_2_CARRY_5,
    PARAM_MESSAGE_2_CARRY_6,
    PARAM_MESSAGE_3_CARRY_3,
    PARAM_MESSAGE_3_CARRY_4,
    PARAM_MESSAGE_3_CARRY_5,
    PARAM_MESSAGE_4_CARRY_4,
)

# concrete-boolean/src/parameters/mod.rs:83-93
DEFAULT_PARAMETERS = BooleanParameters(586, 2, 512, 0.0000925119974676756, 0.0000000342338787018369, 8, 2, 2, 5)
# concrete-boolean/src/parameters/mod.rs:101-111
TFHE_LIB_PARAMETERS = BooleanParameters(630, 1, 1024, 0.0000431583728751555, 0.0000000342338787018369, 7, 3, 2, 8)

SHORTINT_BY_NAME = {name: value for name, value in list(globals().items()) if name.startswith("PARAM_MESSAGE_")}
