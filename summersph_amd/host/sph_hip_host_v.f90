! sph_hip_host_v.f90 -- thin Fortran host over the MI355X SPH core, variable-smoothing-length variant.
!
! Keeps the public face of "SUMMER_SPH - Variable.f90" so that its users can switch:
!   types   particle (with s_length, omega), sink, param     (Variable.f90:14-50)
!   ingest  read_data_from_file: 10 columns x y z vx vy vz u m alpha h, u == 0 marks a sink of
!           radius 5 (:724-843);  read_params_from_file: one header line, then
!           bounding_size max_depth theta gamma eta convergence_criteria max_length timestep_scale end_time (:845-919)
!   loop    simulate(bodies, sinks, params)                    (same step sequence, :1085-1165)
!   saves   make_save: the 10 columns, sinks as x y z vx vy vz 0 m (:921-940)
! The device runs the whole loop body: density with Omega, EOS, Barnes-Hut self-gravity, sink gravity, grad-h SPH
! forces, kicks, drift, dt control, calc_smoothing, check_sink_creation (sinks may be added), sink accretion and the
! boundary cull of gas and sinks.  Not emulated (DESIGN.md): max_depth (the octree here has 21 levels).
module sph_hip_host_v
  use, intrinsic :: iso_c_binding
  use sph_hip_binding
  use sph_hip_textio
  implicit none
  private
  public :: dp, particle, sink, param, read_data_from_file, read_params_from_file, default_params, simulate, make_save

  integer, parameter :: dp = kind(1.0d0)

  type :: particle
    integer :: number
    real(dp) :: mass
    real(dp) :: density
    real(dp) :: internal_energy
    real(dp) :: pressure
    real(dp) :: sound_speed
    real(dp) :: internal_energy_rate
    real(dp) :: alpha
    real(dp) :: alpha_rate
    real(dp) :: s_length
    real(dp) :: omega
    real(dp), dimension(3) :: position
    real(dp), dimension(3) :: velocity
    real(dp), dimension(3) :: acceleration
  end type particle

  type :: sink
    real(dp) :: mass
    real(dp) :: radius
    real(dp), dimension(3) :: spin
    real(dp), dimension(3) :: position
    real(dp), dimension(3) :: velocity
    real(dp), dimension(3) :: acceleration
  end type sink

  type :: param
    real(dp) :: bounding_size
    integer :: max_depth
    real(dp) :: theta
    real(dp) :: gamma
    real(dp) :: eta
    real(dp) :: convergence_criteria
    real(dp) :: max_length
    real(dp) :: timestep_scale
    real(dp) :: end_time
  end type param

contains

  ! The reference ships no parameters.txt; these are the values SURVEY.md 8(d) settles on.
  subroutine default_params(params)
    type(param), intent(out) :: params
    params%bounding_size = 1500.0_dp
    params%max_depth = 1000
    params%theta = 0.5_dp
    params%gamma = 1.4_dp
    params%eta = 1.2_dp
    params%convergence_criteria = 1.0e-3_dp
    params%max_length = 10.0_dp
    params%timestep_scale = 0.25_dp
    params%end_time = 1000.0_dp
  end subroutine default_params

  subroutine read_params_from_file(filename, params)
    character(len=*), intent(in) :: filename
    type(param), intent(inout) :: params
    character(len=1024) :: line
    integer :: unit_no, ios, nrec
    type(param) :: p

    open(newunit=unit_no, file=filename, status='old', action='read', iostat=ios)
    if (ios /= 0) then
      write(*, *) 'Error opening file: ', trim(filename)
      return
    end if
    read(unit_no, '(A)', iostat=ios) line           ! header
    nrec = 0
    do                                               ! the last complete record wins, as in the reference
      read(unit_no, '(A)', iostat=ios) line
      if (ios /= 0) exit
      if (len_trim(line) == 0) cycle
      read(line, *, iostat=ios) p%bounding_size, p%max_depth, p%theta, p%gamma, p%eta, p%convergence_criteria, &
                                p%max_length, p%timestep_scale, p%end_time
      if (ios /= 0) then
        write(*, *) 'Error reading line ', nrec + 1
        exit
      end if
      nrec = nrec + 1
      params = p
    end do
    close(unit_no)
    if (nrec == 0) then
      write(*, *) 'No data found in file: ', trim(filename)
      return
    end if
    write(*, *) 'Successfully read parameters from', trim(filename), '.'
  end subroutine read_params_from_file

  subroutine read_data_from_file(filename, bodies, sinks)
    character(len=*), intent(in) :: filename
    type(particle), allocatable, intent(inout) :: bodies(:)
    type(sink), allocatable, intent(inout) :: sinks(:)
    real(dp), allocatable :: rec(:, :), grown(:, :)
    character(len=1024) :: line
    integer :: unit_no, ios, nrec, ngas, nsink, k, ig, is, pos, nread, nlines
    real(dp) :: v(10)

    open(newunit=unit_no, file=filename, status='old', action='read', iostat=ios)
    if (ios /= 0) then
      write(*, *) 'Error opening file: ', trim(filename)
      return
    end if
    read(unit_no, '(A)', iostat=ios) line          ! header
    allocate(rec(10, 4096))
    nrec = 0
    pos = len(line) + 1                            ! nothing buffered yet
    do
      ! A record is a run of values that may continue over line breaks, as for the reference's list-directed
      ! read ([V]:782; flang wraps list-directed saves at 80 columns): 8 values, and for gas rows (energy /= 0) two more
      ! (alpha, smoothing length); what is left of the record's last line is skipped.  Sink rows carry 8 values
      ! ([V]:938), which the reference's own 10-value read would mis-parse.
      v = 0.0_dp
      call next_values(unit_no, line, pos, v(1:8), nread, nlines, ios)
      if (nread == 0) exit                         ! end of file between records
      if (ios == 0 .and. v(7) /= 0.0_dp) call next_values(unit_no, line, pos, v(9:10), nread, nlines, ios)
      if (ios /= 0) then
        write(*, *) 'Error reading line ', nrec + 1
        exit
      end if
      pos = len(line) + 1                          ! rest of the line belongs to this record
      if (nrec == size(rec, 2)) then
        allocate(grown(10, 2 * nrec))
        grown(:, 1:nrec) = rec
        call move_alloc(grown, rec)
      end if
      nrec = nrec + 1
      rec(:, nrec) = v
    end do
    close(unit_no)
    if (nrec == 0) then
      write(*, *) 'No data found in file: ', trim(filename)
      return
    end if

    nsink = count(rec(7, 1:nrec) == 0.0_dp)
    ngas = nrec - nsink
    if (allocated(bodies)) deallocate(bodies)
    if (allocated(sinks)) deallocate(sinks)
    allocate(bodies(ngas), sinks(max(nsink, 1)))
    ig = 0
    is = 0
    do k = 1, nrec
      if (rec(7, k) /= 0.0_dp) then
        ig = ig + 1
        bodies(ig)%position = rec(1:3, k)
        bodies(ig)%velocity = rec(4:6, k)
        bodies(ig)%internal_energy = rec(7, k)
        bodies(ig)%mass = rec(8, k)
        bodies(ig)%alpha = rec(9, k)               ! read back from saves, Variable.f90:816
        bodies(ig)%alpha_rate = 0.0_dp
        bodies(ig)%s_length = rec(10, k)
        bodies(ig)%omega = 1.0_dp
        bodies(ig)%number = ig
        bodies(ig)%density = 0.0_dp
        bodies(ig)%pressure = 0.0_dp
        bodies(ig)%sound_speed = 0.0_dp
        bodies(ig)%internal_energy_rate = 0.0_dp
        bodies(ig)%acceleration = 0.0_dp
      else
        is = is + 1
        sinks(is)%position = rec(1:3, k)
        sinks(is)%velocity = rec(4:6, k)
        sinks(is)%mass = rec(8, k)
        sinks(is)%radius = 5.0_dp
        sinks(is)%spin = 0.0_dp
        sinks(is)%acceleration = 0.0_dp
      end if
    end do
    if (nsink == 0) then
      sinks(1)%position = 0.0_dp
      sinks(1)%velocity = 0.0_dp
      sinks(1)%acceleration = 0.0_dp
      sinks(1)%spin = 0.0_dp
      sinks(1)%mass = 0.0_dp
      sinks(1)%radius = 0.0_dp
    end if
    write(*, *) 'Successfully read ', size(bodies), ' bodies and ', size(sinks), ' sinks from ', trim(filename), '.'
  end subroutine read_data_from_file

  ! one record per line, explicit format (see sph_hip_host.f90), an existing file is replaced
  subroutine make_save(bodies, sinks, number, filename)
    type(particle), intent(in) :: bodies(:)
    type(sink), intent(in) :: sinks(:)
    integer, intent(in) :: number
    character(len=*), intent(in), optional :: filename
    character(len=256) :: savename
    integer :: io, i

    if (present(filename)) then
      savename = filename
    else
      write(savename, '(A,I0,A)') 'save', number, '.txt'
    end if
    open(newunit=io, file=trim(savename), status='replace', action='write')
    write(io, '(A)') 'x  y  z  vx  vy  vz  energy  mass  alpha  smoothing'
    do i = 1, size(bodies)
      write(io, '(10(1X,ES25.17E3))') bodies(i)%position, bodies(i)%velocity, bodies(i)%internal_energy, &
        bodies(i)%mass, bodies(i)%alpha, bodies(i)%s_length
    end do
    do i = 1, size(sinks)
      write(io, '(8(1X,ES25.17E3))') sinks(i)%position, sinks(i)%velocity, 0.0_dp, sinks(i)%mass
    end do
    close(io)
  end subroutine make_save

  subroutine check(ctx, status, what)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: what
    if (status /= SPH_OK) then
      write(*, *) 'summersph: ', what, ' failed: ', c_message(sph_strerror(status)), ' -- ', c_message(sph_last_error(ctx))
      error stop 1
    end if
  end subroutine check

  subroutine push_state(ctx, bodies, sinks)
    type(c_ptr), intent(in) :: ctx
    type(particle), intent(in) :: bodies(:)
    type(sink), intent(in) :: sinks(:)
    real(c_double), allocatable :: a(:, :), s(:, :)
    integer :: n, ns
    n = size(bodies)
    ns = size(sinks)
    allocate(a(max(n, 1), 10), s(ns, 7))
    a(1:n, 1) = bodies%position(1); a(1:n, 2) = bodies%position(2); a(1:n, 3) = bodies%position(3)
    a(1:n, 4) = bodies%velocity(1); a(1:n, 5) = bodies%velocity(2); a(1:n, 6) = bodies%velocity(3)
    a(1:n, 7) = bodies%internal_energy; a(1:n, 8) = bodies%mass; a(1:n, 9) = bodies%alpha
    a(1:n, 10) = bodies%s_length
    call check(ctx, sph_upload(ctx, int(n, c_int64_t), a(:, 1), a(:, 2), a(:, 3), a(:, 4), a(:, 5), a(:, 6), &
                               a(:, 7), a(:, 8), a(:, 9)), 'sph_upload')
    if (n > 0) call check(ctx, sph_upload_field(ctx, SPH_F_H, a(:, 10), int(n, c_int64_t)), 'sph_upload_field(h)')
    s(:, 1) = sinks%position(1); s(:, 2) = sinks%position(2); s(:, 3) = sinks%position(3)
    s(:, 4) = sinks%velocity(1); s(:, 5) = sinks%velocity(2); s(:, 6) = sinks%velocity(3)
    s(:, 7) = sinks%mass
    call check(ctx, sph_set_sinks(ctx, int(ns, c_int32_t), s(:, 1), s(:, 2), s(:, 3), s(:, 4), s(:, 5), s(:, 6), s(:, 7)), &
               'sph_set_sinks')
    call check(ctx, sph_set_sink_radii(ctx, int(ns, c_int32_t), sinks%radius), 'sph_set_sink_radii')
  end subroutine push_state

  subroutine pull_state(ctx, bodies, sinks, with_derived)
    type(c_ptr), intent(in) :: ctx
    type(particle), intent(inout) :: bodies(:)
    type(sink), intent(inout) :: sinks(:)
    logical, intent(in) :: with_derived
    real(c_double), allocatable :: a(:, :), s(:, :)
    integer :: n, ns, i
    integer(c_int64_t) :: n64
    n = size(bodies)
    ns = size(sinks)
    n64 = int(n, c_int64_t)
    allocate(a(max(n, 1), 10), s(ns, 10))
    call check(ctx, sph_download_state(ctx, n64, a(:, 1), a(:, 2), a(:, 3), a(:, 4), a(:, 5), a(:, 6), a(:, 7), a(:, 8), a(:, 9)), &
               'sph_download_state')
    if (n > 0) call check(ctx, sph_download_field(ctx, SPH_F_H, a(:, 10), n64), 'download h')
    do i = 1, n
      bodies(i)%position = a(i, 1:3)
      bodies(i)%velocity = a(i, 4:6)
      bodies(i)%internal_energy = a(i, 7)
      bodies(i)%mass = a(i, 8)
      bodies(i)%alpha = a(i, 9)
      bodies(i)%s_length = a(i, 10)
    end do
    if (with_derived .and. n > 0) then
      call check(ctx, sph_download_field(ctx, SPH_F_RHO, a(:, 1), n64), 'download rho'); bodies%density = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_OMEGA, a(:, 1), n64), 'download omega'); bodies%omega = a(1:n, 1)
    end if
    call check(ctx, sph_get_sinks(ctx, int(ns, c_int32_t), s(:, 1), s(:, 2), s(:, 3), s(:, 4), s(:, 5), s(:, 6), s(:, 7), &
                                  s(:, 8), s(:, 9), s(:, 10)), 'sph_get_sinks')
    do i = 1, ns
      sinks(i)%position = s(i, 1:3)
      sinks(i)%velocity = s(i, 4:6)
      sinks(i)%mass = s(i, 7)
      sinks(i)%acceleration = s(i, 8:10)
    end do
    call check(ctx, sph_get_sink_radii(ctx, int(ns, c_int32_t), s(:, 1)), 'sph_get_sink_radii')
    sinks%radius = s(1:ns, 1)
  end subroutine pull_state

  ! The time loop, Variable.f90:1085-1165: per step one sph_step call (density, forces, kick, drift, density,
  ! forces, kick, t += dt, next dt, calc_smoothing, accretion, bounds).  Saves every end_time/1000.
  subroutine simulate(bodies, sinks, params, max_steps, quiet, device, dt_log, sph_only, saves)
    type(particle), allocatable, intent(inout) :: bodies(:)
    type(sink), allocatable, intent(inout) :: sinks(:)
    type(param), intent(in) :: params
    integer, intent(in), optional :: max_steps, device
    logical, intent(in), optional :: quiet
    real(dp), allocatable, intent(out), optional :: dt_log(:)
    logical, intent(in), optional :: sph_only     ! .true.: leave out self-gravity, accretion and the cull
    logical, intent(in), optional :: saves        ! periodic saveN.txt files on / off (default: on unless max_steps is given)

    type(c_ptr) :: ctx
    type(sph_params) :: prm
    real(c_double) :: t, dt
    real(dp) :: next_save
    real(dp), allocatable :: dts(:)
    integer :: step, save_no, step_limit, dev, i
    logical :: talk, do_saves

    step_limit = huge(1)
    if (present(max_steps)) step_limit = max_steps
    talk = .true.
    if (present(quiet)) talk = .not. quiet
    dev = 0
    if (present(device)) dev = device
    do_saves = .not. present(max_steps)          ! a step-limited (test) run writes no saveN.txt unless asked to
    if (present(saves)) do_saves = saves

    call check(c_null_ptr, sph_params_default_variable(prm), 'sph_params_default_variable')
    prm%flags = ior(ior(SPH_FLAG_VARIABLE_H, SPH_FLAG_SINK_CREATION), ior(SPH_FLAG_SELF_GRAVITY, SPH_FLAG_ACCRETE_CULL))
    if (present(sph_only)) then
      if (sph_only) prm%flags = SPH_FLAG_VARIABLE_H
    end if
    prm%bounding_size = params%bounding_size
    prm%theta = params%theta
    prm%gamma = params%gamma
    prm%gamma_m1 = params%gamma - 1.0_dp           ! Variable.f90:509
    prm%eta = params%eta
    prm%h_tol = params%convergence_criteria
    prm%h_max_length = params%max_length
    prm%dt_scale = params%timestep_scale
    call check(c_null_ptr, sph_ctx_create(prm, int(dev, c_int), ctx), 'sph_ctx_create')
    call push_state(ctx, bodies, sinks)

    t = 0.0_c_double
    dt = 1.0e-2_c_double
    next_save = 0.0_dp
    save_no = 0
    step = 0
    allocate(dts(0:min(step_limit, 100000)))
    dts(0) = dt

    do while (t < params%end_time .and. step < step_limit)
      ! save check: the reference compares t > t_list(t_test) with t_list(i) = i*end_time/1000 and t_test starting at 0
      ! (out of bounds: observed to write save0 on the first iteration); one save per iteration at most
      if (do_saves) then
        if (save_no == 0 .or. t > next_save) then
          call pull_state(ctx, bodies, sinks, .false.)
          call make_save(bodies, sinks, save_no)
          save_no = save_no + 1
          next_save = (save_no * params%end_time) / 1000
        end if
      end if
      if (talk) print *, 'SPH Particles:', size(bodies), 'dt :', dt, 'time : ', t

      call check(ctx, sph_step(ctx, dt, t), 'sph_step')
      step = step + 1
      if (step <= ubound(dts, 1)) dts(step) = dt

      if (int(sph_count(ctx)) /= size(bodies)) then          ! accretion / cull on the device
        deallocate(bodies)
        allocate(bodies(int(sph_count(ctx))))
      end if
      if (int(sph_sink_count(ctx)) /= size(sinks)) then      ! check_sink_creation added a sink
        deallocate(sinks)
        allocate(sinks(int(sph_sink_count(ctx))))
        do i = 1, size(sinks)
          sinks(i)%spin = 0.0_dp
        end do
      end if
    end do

    ! after a step the h update has invalidated rho and Omega of the new h: only the state comes back then
    call pull_state(ctx, bodies, sinks, .false.)
    do i = 1, size(bodies)
      bodies(i)%number = i
    end do
    if (present(dt_log)) then
      allocate(dt_log(0:min(step, ubound(dts, 1))))
      dt_log = dts(0:ubound(dt_log, 1))
    end if
    call check(ctx, sph_ctx_destroy(ctx), 'sph_ctx_destroy')
  end subroutine simulate

end module sph_hip_host_v
