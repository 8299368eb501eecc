! sph_hip_halo_binding.f90 -- Fortran face of libsummersph_halo.so (include/summersph_halo.h): the loop body of
! simulate() ([F]:889-916) on several GPUs, one process per GPU, ghost exchange as grouped RCCL send/recv on a second
! stream.  The reference has no counterpart (it is one process); run_sph_hip_mg.f90 shows the call sequence.
module sph_hip_halo_binding
  use, intrinsic :: iso_c_binding
  implicit none
  private
  public :: sph_halo_stats, SPH_HALO_ID_BYTES
  public :: sph_halo_unique_id, sph_halo_create, sph_halo_destroy, sph_halo_last_error, sph_halo_set_slabs
  public :: sph_halo_upload, sph_halo_run, sph_halo_count, sph_halo_download, sph_halo_gather_root
  public :: sph_halo_get_stats, sph_halo_selftest

  integer, parameter :: SPH_HALO_ID_BYTES = 128

  type, bind(C) :: sph_halo_stats
    integer(c_int64_t) :: ghosts, migrated, exchanges, collectives, migrations, host_waits, removed, sinks_created
    integer(c_int64_t) :: let_sent, let_received, let_updates
  end type sph_halo_stats

  interface
    integer(c_int) function sph_halo_unique_id(id) bind(C, name='sph_halo_unique_id')
      import :: c_int, c_int8_t
      integer(c_int8_t), intent(out) :: id(*)
    end function

    integer(c_int) function sph_halo_create(ctx, id, rank, nranks, halo) bind(C, name='sph_halo_create')
      import :: c_int, c_int8_t, c_int32_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int8_t), intent(in) :: id(*)
      integer(c_int32_t), value :: rank, nranks
      type(c_ptr), intent(out) :: halo
    end function

    integer(c_int) function sph_halo_destroy(halo) bind(C, name='sph_halo_destroy')
      import :: c_int, c_ptr
      type(c_ptr), value :: halo
    end function

    type(c_ptr) function sph_halo_last_error(halo) bind(C, name='sph_halo_last_error')
      import :: c_ptr
      type(c_ptr), value :: halo
    end function

    integer(c_int) function sph_halo_set_slabs(halo, edges, migrate_every) bind(C, name='sph_halo_set_slabs')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: halo
      real(c_double), intent(in) :: edges(*)
      integer(c_int32_t), value :: migrate_every
    end function

    integer(c_int) function sph_halo_upload(halo, n, x, y, z, vx, vy, vz, u, m, alpha, gid) bind(C, name='sph_halo_upload')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*)
      integer(c_int64_t), intent(in) :: gid(*)
    end function

    integer(c_int) function sph_halo_run(halo, nsteps, dt, t) bind(C, name='sph_halo_run')
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int32_t), value :: nsteps
      real(c_double), intent(inout) :: dt, t
    end function

    integer(c_int64_t) function sph_halo_count(halo) bind(C, name='sph_halo_count')
      import :: c_int64_t, c_ptr
      type(c_ptr), value :: halo
    end function

    integer(c_int) function sph_halo_download(halo, capacity, x, y, z, vx, vy, vz, u, m, alpha, gid) &
        bind(C, name='sph_halo_download')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int64_t), value :: capacity
      real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*)
      integer(c_int64_t), intent(out) :: gid(*)
    end function

    integer(c_int) function sph_halo_gather_root(halo, root, capacity, n_total, x, y, z, vx, vy, vz, u, m, alpha, gid) &
        bind(C, name='sph_halo_gather_root')
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int32_t), value :: root
      integer(c_int64_t), value :: capacity
      integer(c_int64_t), intent(out) :: n_total
      real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*)
      integer(c_int64_t), intent(out) :: gid(*)
    end function

    ! variable-h contexts ("SUMMER_SPH - Variable.f90"): the same three with the smoothing lengths as a 10th array
    integer(c_int) function sph_halo_upload_v(halo, n, x, y, z, vx, vy, vz, u, m, alpha, hsml, gid) bind(C, name='sph_halo_upload_v')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*), hsml(*)
      integer(c_int64_t), intent(in) :: gid(*)
    end function

    integer(c_int) function sph_halo_download_v(halo, capacity, x, y, z, vx, vy, vz, u, m, alpha, hsml, gid) &
        bind(C, name='sph_halo_download_v')
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int64_t), value :: capacity
      real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*), hsml(*)
      integer(c_int64_t), intent(out) :: gid(*)
    end function

    integer(c_int) function sph_halo_gather_root_v(halo, root, capacity, n_total, x, y, z, vx, vy, vz, u, m, alpha, hsml, gid) &
        bind(C, name='sph_halo_gather_root_v')
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: halo
      integer(c_int32_t), value :: root
      integer(c_int64_t), value :: capacity
      integer(c_int64_t), intent(out) :: n_total
      real(c_double), intent(out) :: x(*), y(*), z(*), vx(*), vy(*), vz(*), u(*), m(*), alpha(*), hsml(*)
      integer(c_int64_t), intent(out) :: gid(*)
    end function

    integer(c_int) function sph_halo_get_stats(halo, stats) bind(C, name='sph_halo_get_stats')
      import :: c_int, c_ptr, sph_halo_stats
      type(c_ptr), value :: halo
      type(sph_halo_stats), intent(out) :: stats
    end function

    integer(c_int) function sph_halo_selftest(halo, count) bind(C, name='sph_halo_selftest')
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: halo
      integer(c_int64_t), value :: count
    end function
  end interface
end module sph_hip_halo_binding
