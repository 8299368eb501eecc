! ref_driver_v.f90 -- TEST INFRASTRUCTURE (oracle side), not product code.
!
! Dump driver for the VARIABLE-h reference ("SUMMER_SPH - Variable.f90", module lines 1-1165,
! cut and compiled unmodified by oracle/build_ref.sh).  Calls the reference's own procedures
! in the order its `simulate` does (Variable.f90:1120-1162) and writes what they computed.
!
! Record format (stream): name char(16) | n int64 | n x real64
!
! usage:
!   ref_driver_v eval <ic10.txt> <out.bin> <gamma> <eta> <tol> <maxlen> <scale>
!   ref_driver_v traj <ic10.txt> <out.bin> <gamma> <eta> <tol> <maxlen> <scale> <nsteps> <sph|full>
!   ref_driver_v kernel <r_h.txt> <out.bin>      (file: count, then "r h" pairs)
!   ref_driver_v time <ic10.txt> <out.bin> <gamma> <eta> <tol> <maxlen> <scale> <nsteps> <sph|full>   as traj, nothing dumped per
!                                                 step; prints the wall time of the step loop (bench.py's cpu_baseline)
program ref_driver_v
  use SPH_routines_module
  implicit none
  character(len=512) :: mode, a1, a2, arg
  real(dp) :: gamma, eta, tol, maxlen, scale
  integer :: ou, nsteps
  character(len=16) :: variant
  logical :: timing_only = .false.

  call get_command_argument(1, mode)
  call get_command_argument(2, a1)
  call get_command_argument(3, a2)
  call init_kernel_table()
  call init_grav_kernel_table()
  open(newunit=ou, file=trim(a2), access='stream', form='unformatted', status='replace')
  if (trim(mode) == 'kernel') then
    call run_kernel(trim(a1))
  else
    call get_command_argument(4, arg); read(arg, *) gamma
    call get_command_argument(5, arg); read(arg, *) eta
    call get_command_argument(6, arg); read(arg, *) tol
    call get_command_argument(7, arg); read(arg, *) maxlen
    call get_command_argument(8, arg); read(arg, *) scale
    if (trim(mode) == 'eval') then
      call run_eval(trim(a1))
    else
      timing_only = trim(mode) == 'time'
      call get_command_argument(9, arg); read(arg, *) nsteps
      call get_command_argument(10, variant)
      call run_traj(trim(a1), nsteps, trim(variant) == 'full')
    end if
  end if
  close(ou)

contains

  subroutine put(name, v)
    character(len=*), intent(in) :: name
    real(dp), intent(in) :: v(:)
    character(len=16) :: tag
    tag = name
    write(ou) tag, int(size(v), 8), v
  end subroutine put

  subroutine put_state(prefix, b)
    character(len=*), intent(in) :: prefix
    type(particle), intent(in) :: b(:)
    call put(prefix//'x',  b%position(1)); call put(prefix//'y',  b%position(2)); call put(prefix//'z',  b%position(3))
    call put(prefix//'vx', b%velocity(1)); call put(prefix//'vy', b%velocity(2)); call put(prefix//'vz', b%velocity(3))
    call put(prefix//'u',  b%internal_energy); call put(prefix//'m', b%mass); call put(prefix//'alpha', b%alpha)
    call put(prefix//'h',  b%s_length)
  end subroutine put_state

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

  ! find_forces (Variable.f90:1022-1033) without the Barnes-Hut gas self-gravity call
  subroutine forces_sph_only(root, b, s)
    type(branch), intent(in) :: root
    type(particle), intent(inout) :: b(:)
    type(sink), intent(inout) :: s(:)
    call zero_rates(s, b)
    call sink_gravforces(b, s)
    call get_SPH(root, b)
  end subroutine forces_sph_only

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
    call put_state('', b)
    call put_sinks('', s)
    allocate(root)
    call create_tree(root, b, 1000)
    call put('root_center', root%center)
    call put('root_size', [root%size])
    call get_density(root, b)
    call get_pressure_and_sound_speed(b, gamma)
    call put('rho', b%density); call put('omega', b%omega); call put('P', b%pressure); call put('c', b%sound_speed)
    call forces_sph_only(root, b, s)
    call put_rates('sph_', b, s)
    dt = 1.0e-2_dp
    call get_next_timestep(b, dt, scale)
    call put('sph_dt', [dt])
    call calc_smoothing(root, b, eta, tol, maxlen)
    call put('sph_hnew', b%s_length)
    deallocate(root)
  end subroutine run_eval

  ! iterations of the body of simulate (Variable.f90:1120-1162); "sph" leaves out BH gravity,
  ! sink creation, accretion and the bounds cull
  subroutine run_traj(icfile, nsteps, full)
    character(len=*), intent(in) :: icfile
    integer, intent(in) :: nsteps
    logical, intent(in) :: full
    type(particle), allocatable :: b(:)
    type(sink), allocatable :: s(:)
    type(branch), allocatable :: root
    real(dp) :: t, dt
    real(dp), allocatable :: dts(:), ns(:)
    integer :: i, k
    integer(8) :: c0, c1, crate
    character(len=8) :: pre

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
      call create_tree(root, b, 1000)
      call get_density(root, b)
      call get_pressure_and_sound_speed(b, gamma)
      if (full) then
        call find_forces(root, b, s)
      else
        call forces_sph_only(root, b, s)
      end if
      call kick(b, s, dt)
      deallocate(root)
      call drift(b, s, dt)
      allocate(root)
      call create_tree(root, b, 1000)
      call get_density(root, b)
      call get_pressure_and_sound_speed(b, gamma)
      if (full) then
        call find_forces(root, b, s)
      else
        call forces_sph_only(root, b, s)
      end if
      call kick(b, s, dt)
      t = t + dt
      call get_next_timestep(b, dt, scale)
      call calc_smoothing(root, b, eta, tol, maxlen)
      if (full) then
        call check_sink_creation(b, s, eta)
        if (any(s%mass > 0.0_dp)) call initiate_sink_accretion(s, b, root)
        call check_bounds(b, s, 1500.0_dp)
      end if
      deallocate(root)
      dts(k) = dt
      ns(k) = real(size(b), dp)
      if (timing_only) cycle
      write(pre, '(A,I0,A)') 's', k, '_'
      call put_state(trim(pre), b)
      call put(trim(pre)//'rho', b%density)
      call put(trim(pre)//'omega', b%omega)
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
    real(dp), allocatable :: r(:), hh(:), w(:), dw(:), gw(:)
    open(newunit=iu, file=rfile, status='old', action='read')
    read(iu, *) n
    allocate(r(n), hh(n), w(n), dw(n), gw(n))
    do i = 1, n
      read(iu, *) r(i), hh(i)
    end do
    close(iu)
    do i = 1, n
      call lookup_kernel(r(i), hh(i), w(i), dw(i))
      call lookup_grav_kernel(r(i), hh(i), gw(i))
    end do
    call put('r', r); call put('h', hh); call put('W', w); call put('dW', dw); call put('gW', gw)
    call put('w_table', w_table); call put('dw_table', dw_table)
    call put('consts', [G, pi, dq, real(nq, dp)])
  end subroutine run_kernel
end program ref_driver_v
