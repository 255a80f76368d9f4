/* so100_model_def.h -- RAW model constants of the so100 arm scene, transcribed number by number
 * from the reference's MJCF.  Plain C (also valid C++/HIP).  Nothing here is derived: every
 * derived quantity (euler->quat, dof_M0, invweight0, actuator kv, ...) is computed from these
 * numbers by the model compilers in oracle/so100_oracle.c (fp64) and csrc/so100_model.hpp.
 *
 * Citations: "arm:" = /root/reference/src/so100_mujoco_rl/envs/model/so_arm100_camera.xml,
 *            "scene:" = /root/reference/src/so100_mujoco_rl/envs/model/env01.xml.
 * tests/test_model_def.py re-parses both XML files (when /root/reference is present) and checks
 * every number below against them.
 *
 * Body order as MuJoCo compiles it (depth first; the attach at scene:24-26 comes first):
 *   0 world, 1 so100_Base (no joint: welded), 2 Rotation_Pitch, 3 Upper_Arm, 4 Lower_Arm,
 *   5 Wrist_Pitch_Roll, 6 Fixed_Jaw, 7 Moving_Jaw, 8 block_a (free joint).
 * The arm is a 6-link serial chain: link k (k=0..5) is body k+2, its parent is link k-1
 * (link 0 hangs off the static Base, which sits at the world origin with identity orientation).
 */
#ifndef SO100_MODEL_DEF_H
#define SO100_MODEL_DEF_H

#define SO100_NLINK 6          /* hinge joints / actuated dofs                       */
#define SO100_NQ    13         /* 6 hinge + free joint (3 pos + 4 quat)               */
#define SO100_NV    12         /* 6 hinge + free joint (3 lin + 3 ang)                */
#define SO100_NU    6

/* ---- arm link frames: body pos / orientation relative to parent ------------------------ */
/* arm:72,79,86,93,100,112.  Orientation is given EITHER as quat (w,x,y,z) OR as euler
 * (intrinsic xyz, radians: compiler angle="radian" arm:2).  ORI_KIND: 0 = quat, 1 = euler.   */
static const double SO100_LINK_POS[SO100_NLINK][3] = {
    { 0.0,     -0.0452,   0.0165 },   /* Rotation_Pitch   arm:72  */
    { 0.0,      0.1025,   0.0306 },   /* Upper_Arm        arm:79  */
    { 0.0,      0.11257,  0.028  },   /* Lower_Arm        arm:86  */
    { 0.0,      0.0052,   0.1349 },   /* Wrist_Pitch_Roll arm:93  */
    { 0.0,     -0.0601,   0.0    },   /* Fixed_Jaw        arm:100 */
    {-0.0202,  -0.0244,   0.0    },   /* Moving_Jaw       arm:112 */
};
static const int SO100_LINK_ORI_KIND[SO100_NLINK] = { 0, 1, 1, 1, 1, 0 };
static const double SO100_LINK_ORI[SO100_NLINK][4] = {
    { 0.707105, 0.707108, 0.0, 0.0 },                       /* quat   arm:72  */
    { 1.57079,  0.0, 0.0, 0.0 },                            /* euler  arm:79  */
    {-1.57079,  0.0, 0.0, 0.0 },                            /* euler  arm:86  */
    {-1.57079,  0.0, 0.0, 0.0 },                            /* euler  arm:93  */
    { 0.0, 1.57079, 0.0, 0.0 },                             /* euler  arm:100 */
    { 1.34924e-11, -3.67321e-06, 1.0, -3.67321e-06 },       /* quat   arm:112 */
};

/* ---- arm link inertials (explicit <inertial>): arm:73-74,80-81,87-88,94-95,101-102,113-114 */
static const double SO100_LINK_IPOS[SO100_NLINK][3] = {
    {-9.07886e-05,  0.0590972,   0.031089     },
    {-1.72052e-05,  0.0701802,   0.00310545   },
    {-0.00339604,   0.00137796,  0.0768007    },
    {-0.00852653,  -0.0352279,  -2.34622e-05  },
    { 0.00552377,  -0.0280167,   0.000483583  },
    {-0.00161745,  -0.0303473,   0.000449646  },
};
static const double SO100_LINK_IQUAT[SO100_NLINK][4] = {   /* (w,x,y,z), normalised by the compiler */
    { 0.363978,   0.441169, -0.623108,   0.533504  },
    { 0.50104,    0.498994, -0.493562,   0.50632   },
    { 0.701995,   0.0787996, 0.0645626,  0.704859  },
    {-0.0522806,  0.705235,  0.0549524,  0.704905  },
    { 0.41836,    0.620891, -0.350644,   0.562599  },
    { 0.696562,   0.716737, -0.0239844, -0.0227026 },
};
static const double SO100_LINK_MASS[SO100_NLINK] = {
    0.119226, 0.162409, 0.147968, 0.0661321, 0.0929859, 0.0202444
};
static const double SO100_LINK_DIAGINERTIA[SO100_NLINK][3] = {
    { 5.94278e-05, 5.89975e-05, 3.13712e-05 },
    { 0.000213312, 0.000167164, 7.01522e-05 },
    { 0.000138803, 0.000107748, 4.84242e-05 },
    { 3.45403e-05, 2.39041e-05, 1.94704e-05 },
    { 5.03136e-05, 4.64098e-05, 2.72961e-05 },
    { 1.11265e-05, 8.99651e-06, 2.99548e-06 },
};

/* ---- joints: axis in the link frame, range; joint pos = 0 (anchor at the link origin) ---- */
/* classes arm:34-51; joint elements arm:75,82,89,96,103,115.                                */
static const double SO100_JNT_AXIS[SO100_NLINK][3] = {
    { 0, 1, 0 },    /* Rotation     arm:35 */
    { 1, 0, 0 },    /* Pitch        arm:38 */
    { 1, 0, 0 },    /* Elbow        arm:41 */
    { 1, 0, 0 },    /* Wrist_Pitch  arm:44 */
    { 0, 1, 0 },    /* Wrist_Roll   arm:47 */
    { 0, 0, 1 },    /* Jaw          arm:50 */
};
static const double SO100_JNT_RANGE[SO100_NLINK][2] = {
    {-2.2,      2.2     },
    {-3.14158,  0.2     },
    { 0.0,      3.14158 },
    {-2.0,      1.8     },
    {-3.14158,  3.14158 },
    {-0.2,      2.0     },
};
#define SO100_JNT_FRICTIONLOSS 0.1      /* arm:32 */
#define SO100_JNT_ARMATURE     0.1      /* arm:32 */

/* ---- position actuators: arm:33, arm:139-146 (one per joint, gear 1) --------------------- */
#define SO100_ACT_KP         50.0
#define SO100_ACT_DAMPRATIO  1.0
#define SO100_ACT_FORCE_LO  (-35.0)
#define SO100_ACT_FORCE_HI   35.0
#define SO100_ACT_CTRL_LO   (-3.14158)
#define SO100_ACT_CTRL_HI    3.14158

/* ---- camera on Fixed_Jaw (link 4): arm:125 ---------------------------------------------- */
static const double SO100_CAM_POS[3]   = { -0.001, -0.023827, 0.05778 };
static const double SO100_CAM_EULER[3] = { 4.974, 0.0, 3.142 };
#define SO100_CAM_FOVY_DEG 120.0
#define SO100_CAM_LINK 4

/* ---- finger pads: the 8 box geoms of class "finger_collision" (arm:60-62): arm:108-111 on Fixed_Jaw (link 4),
 * arm:120-123 on Moving_Jaw (link 5).  size = half extents, pos in the link frame, orientation identity.  They are the only
 * arm collision geometry that is not a mesh (the STL meshes of class "collision", arm:58-59, are not in the reference
 * snapshot).  In the reference scene they collide with the floor plane (scene:39; default contype/conaffinity 1) but not
 * with each other (Moving_Jaw is Fixed_Jaw's child: parent-child filter) and not with the cube (excluded, scene:47-48:
 * BASELINE.json configs[4] lifts exactly these two exclusions).                                                        */
#define SO100_NPAD 8
static const int SO100_PAD_LINK[SO100_NPAD] = { 4, 4, 4, 4, 5, 5, 5, 5 };
static const double SO100_PAD_POS[SO100_NPAD][3] = {
    { 0.0089, -0.1014, 0.0 },   /* fixed_jaw_pad_1   arm:108 */
    { 0.0109, -0.0914, 0.0 },   /* fixed_jaw_pad_2   arm:109 */
    { 0.0126, -0.0768, 0.0 },   /* fixed_jaw_pad_3   arm:110 */
    { 0.0143, -0.0572, 0.0 },   /* fixed_jaw_pad_4   arm:111 */
    {-0.0113, -0.077,  0.0 },   /* moving_jaw_pad_1  arm:120 */
    {-0.0093, -0.067,  0.0 },   /* moving_jaw_pad_2  arm:121 */
    {-0.0073, -0.055,  0.0 },   /* moving_jaw_pad_3  arm:122 */
    {-0.0073, -0.035,  0.0 },   /* moving_jaw_pad_4  arm:123 */
};
static const double SO100_PAD_SIZE[SO100_NPAD][3] = {
    { 0.001, 0.005, 0.004 },
    { 0.001, 0.005, 0.006 },
    { 0.001, 0.01,  0.007 },
    { 0.001, 0.01,  0.008 },
    { 0.001, 0.005, 0.004 },
    { 0.001, 0.005, 0.006 },
    { 0.001, 0.01,  0.006 },
    { 0.001, 0.01,  0.008 },
};
/* arm:61  solimp="2 1 0.01" solref="0.01 1" friction="1 0.005 0.0001" (midpoint / power stay at their defaults) */
#define SO100_PAD_SOLREF_TIMECONST 0.01
#define SO100_PAD_SOLREF_DAMPRATIO 1.0
#define SO100_PAD_SOLIMP_D0     2.0
#define SO100_PAD_SOLIMP_DMAX   1.0
#define SO100_PAD_SOLIMP_WIDTH  0.01
#define SO100_PAD_FRICTION      1.0
/* mj_assignImp clamps d0, dmax (and the midpoint) into [mjMINIMP, mjMAXIMP] AFTER the two geoms' solimp have been mixed */
#define SO100_MJMINIMP 0.0001
#define SO100_MJMAXIMP 0.9999

/* ---- cube "block_a": scene:29-35.  inertiafromgeom="true" (scene:2) => mass and inertia come
 * from the box geom (half size 0.01, default density 1000), COM at the geom centre (0,0,0);
 * the <inertial mass="0.2"> at scene:33 is overridden.              */
#define SO100_CUBE_HALF      0.01
#define SO100_GEOM_DENSITY   1000.0     /* MuJoCo default geom density */

/* ---- mjOption defaults in force (scene:3 override is commented out) ------- */
#define SO100_TIMESTEP   0.002
#define SO100_GRAVITY_Z (-9.81)
/* default solref / solimp for joint limits, friction loss, contacts */
#define SO100_SOLREF_TIMECONST 0.02
#define SO100_SOLREF_DAMPRATIO 1.0
#define SO100_SOLIMP_D0     0.9
#define SO100_SOLIMP_DMAX   0.95
#define SO100_SOLIMP_WIDTH  0.001
#define SO100_SOLIMP_MID    0.5
#define SO100_SOLIMP_POWER  2.0
#define SO100_MJMINVAL      1e-15
/* default geom friction (sliding) used by the cube-floor contact */
#define SO100_GEOM_FRICTION 1.0

/* ---- link proxies: STAND-IN geometry, NOT numbers of the reference -----------------------------------------------------------
 * The arm's collision geoms other than the finger pads are MESHES (class "collision", arm:58-59, used at arm:85, 92, 99, 106-107,
 * 117-119) whose STL files are not in the reference snapshot.  Behind their own physics flag (SO100_F_LINKS_FLOOR) each of these
 * links is given ONE capsule, built by rule from numbers that ARE in the MJCF:
 *   links 1-3 (Upper_Arm, Lower_Arm, Wrist_Pitch_Roll): segment from the link's own joint origin to its child's joint origin
 *     (SO100_LINK_POS of the child, arm:86, 93, 100);
 *   links 4-5 (Fixed_Jaw, Moving_Jaw): segment from the jaw's joint origin to the far end of the box around its finger pads
 *     (centre line of the pads in x, their largest |y|, z = 0; arm:108-111, 120-123);
 *   radius = mean of the two SMALLER half sizes of the solid box that has the link's mass and principal inertias
 *     (sx^2 = 6 (I_y + I_z - I_x) / m, ...; arm:73-74 ... 113-114), for the jaws capped at SO100_PROX_JAW_RADIUS_MAX.
 * Contact parameters: MuJoCo's defaults (the meshes' class sets none).  Parity: unpinned by construction. */
#define SO100_NPROX 5
static const int SO100_PROX_LINK[SO100_NPROX] = { 1, 2, 3, 4, 5 };
/* SO100_F_LINKS_CUBE: the same rule on links 0 (Rotation_Pitch) and 1 (Upper_Arm) -- segment from the link's joint origin to its child's, radius from
 * the inertia box -- against block_a: the two arm bodies that scene:44-48 does NOT exclude from colliding with the cube. */
#define SO100_NCPROX 2
#define SO100_PROX_JAW_RADIUS_MAX 0.008   /* a jaw is a thin finger (pads: 8 mm half height): its inertia box -- it carries the servo -- would give a 2 cm capsule that buries the pads */

#endif /* SO100_MODEL_DEF_H */
