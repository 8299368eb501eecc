! sph_hip_binding.f90 -- ISO_C_BINDING view of include/summersph.h (libsummersph_hip.so).
!
! This is the stub a maintainer of the reference adds to call the MI355X core from
! Fortran: one `bind(C)` interface per C entry point, plus the parameter struct.  Each
! interface names the reference procedure it stands in for (file SUMMER_SPH.f90).
module sph_hip_binding
  use, intrinsic :: iso_c_binding
  implicit none
  private
  public :: sph_params, sph_stats
  public :: sph_params_default, sph_ctx_create, sph_ctx_destroy, sph_strerror, sph_last_error, sph_abi_version, sph_get_params
  public :: sph_upload, sph_set_sinks, sph_get_sinks, sph_count
  public :: sph_density, sph_forces, sph_kick, sph_drift, sph_next_dt, sph_step, sph_run
  public :: sph_download_field, sph_download_state, sph_get_stats, sph_get_bbox, sph_synchronize
  public :: SPH_OK, SPH_F_X, SPH_F_Y, SPH_F_Z, SPH_F_VX, SPH_F_VY, SPH_F_VZ, SPH_F_U, SPH_F_M, SPH_F_ALPHA
  public :: SPH_F_RHO, SPH_F_P, SPH_F_C, SPH_F_AX, SPH_F_AY, SPH_F_AZ, SPH_F_DU, SPH_F_DALPHA
  public :: sph_set_sink_radii, sph_accrete_and_cull, sph_upload_field, sph_update_h, sph_params_default_variable
  public :: SPH_F_H, SPH_F_OMEGA
  public :: SPH_FLAG_REUSE_DENSITY, SPH_FLAG_VARIABLE_H, SPH_FLAG_SELF_GRAVITY, SPH_FLAG_ACCRETE_CULL
  public :: SPH_FLAG_SINK_CREATION, sph_sink_count, sph_get_sink_radii
  ! multi-GPU building blocks (device pointers as type(c_ptr), e.g. from hipMalloc or an MPI library's GPU buffers)
  public :: sph_set_owned, sph_set_rank, sph_reserve, sph_owned_bbox, sph_select_boxes, sph_selected_ids_dev
  public :: sph_select_boxes_async, sph_selected_counts, sph_gather_selected_dev
  public :: sph_replace_ghosts_dev, sph_gather_fields_dev, sph_scatter_fields_dev, sph_refresh_eos_ghosts
  public :: sph_set_boundary_boxes, sph_forces_part, sph_set_dt, sph_get_dt, sph_kick_devdt, sph_drift_devdt
  public :: sph_kick_drift_devdt, sph_kick_dt_candidate_dev, sph_kick_dt_candidate_gas_dev, sph_kick_sinks_devdt
  public :: sph_dt_candidate_dev, sph_pack_partials_dev, sph_pack_partials_ex_dev, sph_apply_partials_dev, sph_set_gravity_sources_dev
  public :: SPH_PARTIALS
  public :: c_message

  integer(c_int), parameter :: SPH_OK = 0
  integer(c_int), parameter :: SPH_F_X = 0, SPH_F_Y = 1, SPH_F_Z = 2, SPH_F_VX = 3, SPH_F_VY = 4, SPH_F_VZ = 5
  integer(c_int), parameter :: SPH_F_U = 6, SPH_F_M = 7, SPH_F_ALPHA = 8, SPH_F_RHO = 9, SPH_F_P = 10, SPH_F_C = 11
  integer(c_int), parameter :: SPH_F_AX = 12, SPH_F_AY = 13, SPH_F_AZ = 14, SPH_F_DU = 15, SPH_F_DALPHA = 16

  integer(c_int), parameter :: SPH_F_H = 17, SPH_F_OMEGA = 18
  integer(c_int32_t), parameter :: SPH_FLAG_REUSE_DENSITY = 1, SPH_FLAG_VARIABLE_H = 2, SPH_FLAG_SELF_GRAVITY = 16
  integer(c_int32_t), parameter :: SPH_FLAG_ACCRETE_CULL = 32, SPH_FLAG_SINK_CREATION = 64
  integer(c_int32_t), parameter :: SPH_PARTIALS = 199

  type, bind(C) :: sph_params
    real(c_double) :: h, gamma, gamma_m1
    integer(c_int32_t) :: nq, flags
    real(c_double) :: kernel_pi, visc_eps, alpha_floor, alpha_decay, G, dt_scale, dt_max, dt_min, bounding_size
    real(c_double) :: eta, h_tol, h_max_length, h_min_length, h_iter_cap     ! variable-h path only
    real(c_double) :: theta                                                   ! self-gravity opening angle
  end type sph_params

  type, bind(C) :: sph_stats
    integer(c_int64_t) :: n, n_cells
    integer(c_int32_t) :: grid_dim(3)
    integer(c_int32_t) :: nlist_capacity, nlist_max, tile_fit_pct
    real(c_double) :: nlist_mean
    integer(c_int64_t) :: grid_builds, nlist_builds, density_passes, force_passes, device_bytes
    real(c_double) :: nlist_wave_mean
    integer(c_int32_t) :: tile_fit_pct_forces, host_syncs
    real(c_double) :: lane_efficiency_forces
    integer(c_int64_t) :: nlist_reflags
  end type sph_stats

  interface
    integer(c_int) function sph_abi_version() bind(C, name='sph_abi_version')
      import :: c_int
    end function

    integer(c_int) function sph_get_params(ctx, p) bind(C, name='sph_get_params')
      import :: c_int, c_ptr, sph_params
      type(c_ptr), value :: ctx
      type(sph_params), intent(out) :: p
    end function

    integer(c_int) function sph_params_default(p) bind(C, name='sph_params_default')
      import :: c_int, sph_params
      type(sph_params), intent(out) :: p
    end function

    ! stands in for init_kernel_table (:55-79) and the per-step tree allocation (:894,905)
    integer(c_int) function sph_ctx_create(p, device, ctx) bind(C, name='sph_ctx_create')
      import :: c_int, c_ptr, sph_params
      type(sph_params), intent(in) :: p
      integer(c_int), value :: device
      type(c_ptr), intent(out) :: ctx
    end function

    integer(c_int) function sph_ctx_destroy(ctx) bind(C, name='sph_ctx_destroy')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    type(c_ptr) function sph_strerror(status) bind(C, name='sph_strerror')
      import :: c_int, c_ptr
      integer(c_int), value :: status
    end function

    type(c_ptr) function sph_last_error(ctx) bind(C, name='sph_last_error')
      import :: c_ptr
      type(c_ptr), value :: ctx
    end function

    ! hand-over of `type(particle) :: bodies(:)` as struct-of-arrays (:14-27)
    integer(c_int) function sph_upload(ctx, n, x, y, z, vx, vy, vz, u, m, alpha) bind(C, name='sph_upload')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*)
    end function

    ! hand-over of `type(sink) :: sinks(:)` (:30-37)
    integer(c_int) function sph_set_sinks(ctx, ns, sx, sy, sz, svx, svy, svz, sm) bind(C, name='sph_set_sinks')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: ns
      real(c_double), intent(in) :: sx(*), sy(*), sz(*), svx(*), svy(*), svz(*), sm(*)
    end function

    integer(c_int) function sph_get_sinks(ctx, ns, sx, sy, sz, svx, svy, svz, sm, sax, say, saz) &
        bind(C, name='sph_get_sinks')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: ns
      real(c_double), intent(out) :: sx(*), sy(*), sz(*), svx(*), svy(*), svz(*), sm(*), sax(*), say(*), saz(*)
    end function

    integer(c_int) function sph_params_default_variable(p) bind(C, name='sph_params_default_variable')
      import :: c_int, sph_params
      type(sph_params), intent(out) :: p
    end function

    ! sink%radius (:694)
    integer(c_int) function sph_set_sink_radii(ctx, ns, radius) bind(C, name='sph_set_sink_radii')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: ns
      real(c_double), intent(in) :: radius(*)
    end function

    ! initiate_sink_accretion + check_bounds (:919-920)
    integer(c_int) function sph_accrete_and_cull(ctx, n_removed) bind(C, name='sph_accrete_and_cull')
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int64_t), intent(out) :: n_removed
    end function

    ! one field in the caller's particle order (e.g. the smoothing lengths of the variable-h variant)
    integer(c_int) function sph_upload_field(ctx, field, host, n) bind(C, name='sph_upload_field')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: field
      real(c_double), intent(in) :: host(*)
      integer(c_int64_t), value :: n
    end function

    ! calc_smoothing of the variable-h variant
    integer(c_int) function sph_update_h(ctx) bind(C, name='sph_update_h')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int64_t) function sph_count(ctx) bind(C, name='sph_count')
      import :: c_int64_t, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! number of sinks (check_sink_creation of the variable-h variant may add one per step)
    integer(c_int32_t) function sph_sink_count(ctx) bind(C, name='sph_sink_count')
      import :: c_int32_t, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function sph_get_sink_radii(ctx, ns, radius) bind(C, name='sph_get_sink_radii')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: ns
      real(c_double), intent(out) :: radius(*)
    end function

    ! create_tree + get_density + get_pressure_and_sound_speed (:894-897)
    integer(c_int) function sph_density(ctx) bind(C, name='sph_density')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! find_forces (:818-829) without the Barnes-Hut gas self-gravity term
    integer(c_int) function sph_forces(ctx) bind(C, name='sph_forces')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! kick (:742-759)
    integer(c_int) function sph_kick(ctx, dt) bind(C, name='sph_kick')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), value :: dt
    end function

    ! drift (:762-776)
    integer(c_int) function sph_drift(ctx, dt) bind(C, name='sph_drift')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), value :: dt
    end function

    ! get_next_timestep (:831-860)
    integer(c_int) function sph_next_dt(ctx, dt) bind(C, name='sph_next_dt')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: dt
    end function

    ! one iteration of simulate's loop body (:889-916)
    integer(c_int) function sph_step(ctx, dt, t) bind(C, name='sph_step')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: dt, t
    end function

    integer(c_int) function sph_run(ctx, nsteps, dt, t) bind(C, name='sph_run')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nsteps
      real(c_double), intent(inout) :: dt, t
    end function

    integer(c_int) function sph_download_field(ctx, field, host, n) bind(C, name='sph_download_field')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: field
      real(c_double), intent(out) :: host(*)
      integer(c_int64_t), value :: n
    end function

    integer(c_int) function sph_download_state(ctx, n, x, y, z, vx, vy, vz, u, m, alpha) &
        bind(C, name='sph_download_state')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: n
      real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*)
    end function

    integer(c_int) function sph_get_stats(ctx, st) bind(C, name='sph_get_stats')
      import :: c_int, c_ptr, sph_stats
      type(c_ptr), value :: ctx
      type(sph_stats), intent(out) :: st
    end function

    ! bounding box of the current positions: what check_bounds (:471-482) needs
    integer(c_int) function sph_get_bbox(ctx, lo, hi) bind(C, name='sph_get_bbox')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: lo(3), hi(3)
    end function

    integer(c_int) function sph_synchronize(ctx) bind(C, name='sph_synchronize')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! ---- one context per GPU: owned particles + ghost copies, exchanged by the caller (MPI, RCCL, ...) -------------
    integer(c_int) function sph_set_owned(ctx, n_owned) bind(C, name='sph_set_owned')
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: n_owned
    end function
    integer(c_int) function sph_set_rank(ctx, rank, nranks) bind(C, name='sph_set_rank')
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: rank, nranks
    end function
    integer(c_int) function sph_reserve(ctx, n_slots) bind(C, name='sph_reserve')
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: n_slots
    end function
    integer(c_int) function sph_owned_bbox(ctx, lo_hi, d_lo_hi) bind(C, name='sph_owned_bbox')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, d_lo_hi
      real(c_double), intent(out) :: lo_hi(6)
    end function
    integer(c_int) function sph_select_boxes(ctx, nbox, boxes, counts) bind(C, name='sph_select_boxes')
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nbox
      real(c_double), intent(in) :: boxes(6, *)
      integer(c_int64_t), intent(out) :: counts(*)
    end function
    integer(c_int) function sph_selected_ids_dev(ctx, box, count, d_ids) bind(C, name='sph_selected_ids_dev')
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: ctx, d_ids
      integer(c_int32_t), value :: box
      integer(c_int64_t), value :: count
    end function
    integer(c_int) function sph_select_boxes_async(ctx, nbox, boxes) bind(C, name='sph_select_boxes_async')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nbox
      real(c_double), intent(in) :: boxes(*)
    end function
    integer(c_int) function sph_selected_counts(ctx, nbox, counts) bind(C, name='sph_selected_counts')
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nbox
      integer(c_int64_t), intent(out) :: counts(*)
    end function
    integer(c_int) function sph_gather_selected_dev(ctx, box, nf, fields, capacity, d_out) bind(C, name='sph_gather_selected_dev')
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: ctx, d_out
      integer(c_int32_t), value :: box, nf
      integer(c_int32_t), intent(in) :: fields(*)
      integer(c_int64_t), value :: capacity
    end function
    integer(c_int) function sph_replace_ghosts_dev(ctx, count, d_state) bind(C, name='sph_replace_ghosts_dev')
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ctx, d_state
      integer(c_int64_t), value :: count
    end function
    integer(c_int) function sph_gather_fields_dev(ctx, nf, fields, count, d_ids, d_out) bind(C, name='sph_gather_fields_dev')
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: ctx, d_ids, d_out
      integer(c_int32_t), value :: nf
      integer(c_int32_t), intent(in) :: fields(*)
      integer(c_int64_t), value :: count
    end function
    integer(c_int) function sph_scatter_fields_dev(ctx, nf, fields, first, count, d_vals) bind(C, name='sph_scatter_fields_dev')
      import :: c_int, c_int32_t, c_int64_t, c_ptr
      type(c_ptr), value :: ctx, d_vals
      integer(c_int32_t), value :: nf
      integer(c_int32_t), intent(in) :: fields(*)
      integer(c_int64_t), value :: first, count
    end function
    integer(c_int) function sph_refresh_eos_ghosts(ctx) bind(C, name='sph_refresh_eos_ghosts')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function sph_set_boundary_boxes(ctx, nbox, boxes) bind(C, name='sph_set_boundary_boxes')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nbox
      real(c_double), intent(in) :: boxes(6, *)
    end function
    integer(c_int) function sph_forces_part(ctx, part) bind(C, name='sph_forces_part')
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: part
    end function
    integer(c_int) function sph_set_dt(ctx, dt, t) bind(C, name='sph_set_dt')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), value :: dt, t
    end function
    integer(c_int) function sph_get_dt(ctx, dt, t) bind(C, name='sph_get_dt')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: dt, t
    end function
    integer(c_int) function sph_kick_devdt(ctx) bind(C, name='sph_kick_devdt')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function sph_kick_drift_devdt(ctx) bind(C, name='sph_kick_drift_devdt')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function sph_kick_dt_candidate_gas_dev(ctx) bind(C, name='sph_kick_dt_candidate_gas_dev')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function sph_kick_sinks_devdt(ctx) bind(C, name='sph_kick_sinks_devdt')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function sph_kick_dt_candidate_dev(ctx) bind(C, name='sph_kick_dt_candidate_dev')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function sph_drift_devdt(ctx) bind(C, name='sph_drift_devdt')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function sph_dt_candidate_dev(ctx) bind(C, name='sph_dt_candidate_dev')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function sph_pack_partials_ex_dev(ctx, d_out, predict_box) bind(C, name='sph_pack_partials_ex_dev')
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: ctx, d_out
      integer(c_int32_t), value :: predict_box
    end function

    integer(c_int) function sph_pack_partials_dev(ctx, d_out) bind(C, name='sph_pack_partials_dev')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, d_out
    end function
    integer(c_int) function sph_apply_partials_dev(ctx, d_all, nranks, stride, apply_dt) bind(C, name='sph_apply_partials_dev')
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: ctx, d_all
      integer(c_int32_t), value :: nranks, stride, apply_dt
    end function
    integer(c_int) function sph_set_gravity_sources_dev(ctx, n_src, d_xyzm, lo_hi) bind(C, name='sph_set_gravity_sources_dev')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ctx, d_xyzm
      integer(c_int64_t), value :: n_src
      real(c_double), intent(in) :: lo_hi(6)
    end function
  end interface

contains

  ! copies a NUL-terminated C string into a Fortran string
  function c_message(cp) result(s)
    type(c_ptr), intent(in) :: cp
    character(len=:), allocatable :: s
    character(kind=c_char), pointer :: chars(:)
    integer :: k
    s = ''
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, chars, [4096])
    do k = 1, 4096
      if (chars(k) == c_null_char) exit
      s = s // chars(k)
    end do
  end function c_message
end module sph_hip_binding
