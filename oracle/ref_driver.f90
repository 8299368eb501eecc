! ref_driver.f90 -- TEST INFRASTRUCTURE (oracle side), not product code.
!
! A dump driver that links against the *unmodified* reference module
! `SPH_routines_module` (cut from /root/reference/SUMMER_SPH.f90:1-931 by
! oracle/build_ref.sh at build time; the reference source itself is never
! copied into this repository).  It calls the reference's own procedures in the
! order `simulate` does (SUMMER_SPH.f90:879-929) and writes what they computed
! as raw little-endian records so that golden fixtures can be generated.
!
! Record format (stream): name char(16) | n int64 | n x real64
!
! usage:
!   ref_driver eval   <ic.txt> <out.bin>
!   ref_driver traj   <ic.txt> <out.bin> <nsteps> <sph|full>
!   ref_driver kernel <r.txt>  <out.bin>          (r.txt: count, then r values)
!   ref_driver time   <ic.txt> <out.bin> <nsteps> <sph|full>    as traj, nothing dumped per step; prints the wall time of
!                                                  the step loop (ingest excluded): bench.py's cpu_baseline, kind "reference"
program ref_driver
  use SPH_routines_module
  implicit none
  character(len=512) :: mode, a1, a2, a3, a4
  integer :: ou
  logical :: timing_only = .false.

  call get_command_argument(1, mode)
  call get_command_argument(2, a1)
  call get_command_argument(3, a2)
  call get_command_argument(4, a3)
  call get_command_argument(5, a4)

  call init_kernel_table()
  call init_grav_kernel_table()

  open(newunit=ou, file=trim(a2), access='stream', form='unformatted', status='replace')
  select case (trim(mode))
  case ('eval')
    call run_eval(trim(a1))
  case ('traj')
    call run_traj(trim(a1), trim(a3), trim(a4))
  case ('time')
    timing_only = .true.
    call run_traj(trim(a1), trim(a3), trim(a4))
  case ('kernel')
    call run_kernel(trim(a1))
  case default
    print *, 'unknown mode'
    stop 2
  end select
  close(ou)

contains

  subroutine put(name, v)
    character(len=*), intent(in) :: name
    real(dp), intent(in) :: v(:)
    character(len=16) :: tag
    tag = name
    write(ou) tag, int(size(v), 8), v
  end subroutine put

  subroutine put_gas_state(prefix, b)
    character(len=*), intent(in) :: prefix
    type(particle), intent(in) :: b(:)
    call put(prefix//'x',  b%position(1)); call put(prefix//'y',  b%position(2)); call put(prefix//'z',  b%position(3))
    call put(prefix//'vx', b%velocity(1)); call put(prefix//'vy', b%velocity(2)); call put(prefix//'vz', b%velocity(3))
    call put(prefix//'u',  b%internal_energy); call put(prefix//'m', b%mass); call put(prefix//'alpha', b%alpha)
  end subroutine put_gas_state

  subroutine put_rates(prefix, b, s)
    character(len=*), intent(in) :: prefix
    type(particle), intent(in) :: b(:)
    type(sink), intent(in) :: s(:)
    call put(prefix//'ax', b%acceleration(1)); call put(prefix//'ay', b%acceleration(2)); call put(prefix//'az', b%acceleration(3))
    call put(prefix//'du', b%internal_energy_rate); call put(prefix//'dalpha', b%alpha_rate)
    call put(prefix//'sax', s%acceleration(1)); call put(prefix//'say', s%acceleration(2)); call put(prefix//'saz', s%acceleration(3))
  end subroutine put_rates

  subroutine put_sinks(prefix, s)
    character(len=*), intent(in) :: prefix
    type(sink), intent(in) :: s(:)
    call put(prefix//'sx', s%position(1)); call put(prefix//'sy', s%position(2)); call put(prefix//'sz', s%position(3))
    call put(prefix//'svx', s%velocity(1)); call put(prefix//'svy', s%velocity(2)); call put(prefix//'svz', s%velocity(3))
    call put(prefix//'sm', s%mass); call put(prefix//'srad', s%radius)
  end subroutine put_sinks

  ! find_forces without the Barnes-Hut gas self-gravity term: the same calls as
  ! SUMMER_SPH.f90:824,826,827 (zero_rates, sink_gravforces, get_SPH).
  subroutine forces_sph_only(root, b, s)
    type(branch), intent(in) :: root
    type(particle), intent(inout) :: b(:)
    type(sink), intent(inout) :: s(:)
    call zero_rates(s, b)
    call sink_gravforces(b, s)
    call get_SPH(root, b)
  end subroutine forces_sph_only

  ! one force evaluation on the ingested state, both with and without BH gravity
  subroutine run_eval(icfile)
    character(len=*), intent(in) :: icfile
    type(particle), allocatable :: b(:)
    type(sink), allocatable :: s(:)
    type(branch), allocatable :: root
    real(dp) :: dt
    integer :: i

    call read_data_from_file(icfile, b, s)
    do i = 1, size(b)
      b(i)%number = i
    end do
    call put_gas_state('', b)
    call put_sinks('', s)

    allocate(root)
    call create_tree(root, b, max_depth)
    call put('root_center', root%center)
    call put('root_size', [root%size])
    call get_density(root, b)
    call get_pressure_and_sound_speed(b)
    call put('rho', b%density); call put('P', b%pressure); call put('c', b%sound_speed)

    call forces_sph_only(root, b, s)
    call put_rates('sph_', b, s)
    dt = 1.0e-2_dp
    call get_next_timestep(b, dt)
    call put('sph_dt', [dt])

    call find_forces(root, b, s)
    call put_rates('full_', b, s)
    dt = 1.0e-2_dp
    call get_next_timestep(b, dt)
    call put('full_dt', [dt])
    deallocate(root)
  end subroutine run_eval

  ! nsteps iterations of the body of `simulate` (SUMMER_SPH.f90:886-928)
  subroutine run_traj(icfile, nsteps_s, variant)
    character(len=*), intent(in) :: icfile, nsteps_s, variant
    type(particle), allocatable :: b(:)
    type(sink), allocatable :: s(:)
    type(branch), allocatable :: root
    real(dp) :: t, dt
    real(dp), allocatable :: dts(:), ns(:)
    integer :: i, k, nsteps
    integer(8) :: c0, c1, crate
    logical :: full
    character(len=8) :: pre

    read(nsteps_s, *) nsteps
    full = (variant == 'full')
    call read_data_from_file(icfile, b, s)
    allocate(dts(0:nsteps), ns(0:nsteps))
    t = 0.0_dp
    dt = 1.0e-2_dp
    dts(0) = dt
    ns(0) = real(size(b), dp)

    call system_clock(c0, crate)
    do k = 1, nsteps
      do i = 1, size(b)
        b(i)%number = i
      end do
      allocate(root)
      call create_tree(root, b, max_depth)
      call get_density(root, b)
      call get_pressure_and_sound_speed(b)
      if (full) then
        call find_forces(root, b, s)
      else
        call forces_sph_only(root, b, s)
      end if
      call kick(b, s, dt)
      deallocate(root)
      call drift(b, s, dt)
      allocate(root)
      call create_tree(root, b, max_depth)
      call get_density(root, b)
      call get_pressure_and_sound_speed(b)
      if (full) then
        call find_forces(root, b, s)
      else
        call forces_sph_only(root, b, s)
      end if
      call kick(b, s, dt)
      t = t + dt
      call get_next_timestep(b, dt)
      if (full) then
        if (any(s%mass > 0.0_dp)) call initiate_sink_accretion(s, b, root)
        call check_bounds(b)
      end if
      deallocate(root)
      dts(k) = dt
      ns(k) = real(size(b), dp)
      if (timing_only) cycle
      write(pre, '(A,I0,A)') 's', k, '_'
      call put_gas_state(trim(pre), b)
      call put(trim(pre)//'rho', b%density)
      call put_rates(trim(pre), b, s)
      call put_sinks(trim(pre), s)
    end do
    call system_clock(c1)
    if (timing_only) write(*, '(A,ES16.8,A,I0,A,I0)') 'loop_seconds ', real(c1 - c0, dp) / real(crate, dp), ' particles ', int(ns(0)), ' steps ', nsteps
    call put('dt_seq', dts)
    call put('n_seq', ns)
    call put('t_end', [t])
  end subroutine run_traj

  subroutine run_kernel(rfile)
    character(len=*), intent(in) :: rfile
    integer :: n, i, iu
    real(dp), allocatable :: r(:), w(:), dw(:), gw(:)
    open(newunit=iu, file=rfile, status='old', action='read')
    read(iu, *) n
    allocate(r(n), w(n), dw(n), gw(n))
    do i = 1, n
      read(iu, *) r(i)
    end do
    close(iu)
    do i = 1, n
      call lookup_kernel(r(i), smoothing, w(i), dw(i))
      call lookup_grav_kernel(r(i), smoothing, gw(i))
    end do
    call put('r', r); call put('W', w); call put('dW', dw); call put('gW', gw)
    call put('w_table', w_table); call put('dw_table', dw_table); call put('grav_table', grav_table)
    call put('consts', [G, smoothing, dq, real(nq, dp), bounding_size])
  end subroutine run_kernel
end program ref_driver
